#!/usr/bin/env python3
"""Diagnostic: wave-level counters of the lane-per-member entropy decoder (la_inflate_lanes.hip, LA_DIAG build:
`make -C libarchive_amd/csrc diag`) on the C3 member shape.  Not a benchmark: the counters cost time."""
import ctypes as C, os, sys, zlib, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("LA_GPU_BID", "all")
import numpy as np
import bench as B
import streams as S
from concurrent.futures import ThreadPoolExecutor

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = (mib << 20) // (B.BPF * B.BLOCK)
_, plain = S.synth_lz4_stream(B.SEED, 0, frames, B.BPF, B.BLOCK, nthreads=16)
pieces = [(plain[i:i + B.BLOCK].tobytes(),) for i in range(0, plain.size, B.BLOCK)]
with ThreadPoolExecutor(16) as pool:
    members = list(pool.map(B._gz_make_member, pieces, chunksize=64))
import torch
import libarchive_amd._native as N
N.GPU_LIB_PATH = os.environ.get("LA_DIAG_LIB") or os.path.join(ROOT, "libarchive_amd", "csrc", "libla_gpu_diag.so")
import libarchive_amd as la
from libarchive_amd.gzip import GzDevicePlan
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
img = np.frombuffer(b"".join(members), dtype=np.uint8)
idx = la.gz_index(img, at_eof=True)
d_src = torch.from_numpy(img.copy()).cuda()
plan = GzDevicePlan(ctx, d_src, idx)
plan.run(0); ctx.sync()
cnt = torch.zeros(16, dtype=torch.int64, device="cuda")
assert la.gpu_lib().la_diag_set_il_counters(C.c_void_p(cnt.data_ptr())) == 0
plan.run(0); ctx.sync()
c = cnt.cpu().numpy().astype(np.float64)
nm = len(idx.members)
waves = c[1]
print("members %d, waves %d, compressed %d, decoded %d" % (nm, waves, img.size, plain.size))
print("wave lifetime %.0f cycles (lane 0)" % (c[0] / waves))
print("deflate blocks seen per wave (wave-level header regions) %.1f, header + table build %.0f cycles each = %.1f %% of the lifetime" %
      (c[3] / waves, c[2] / max(c[3], 1), 100 * c[2] / c[0]))
print("outer symbol iterations per wave %.0f (%.0f cycles each incl. everything but headers); match paths per wave %.0f" %
      (c[6] / waves, (c[0] - c[2]) / c[6], c[7] / waves))
print("literal/length long-code walks per wave %.0f (%.2f per outer iteration), lengths tried per walk %.2f" %
      (c[4] / waves, c[4] / c[6], c[5] / max(c[4], 1)))
print("distance long-code walks per wave %.0f (incl. code-length code), lengths tried per walk %.2f" % (c[8] / waves, c[9] / max(c[8], 1)))
print("first-lane samples: sequences per member %.0f, literals per member %.0f" % (c[10] / waves / 1, c[11] / waves / 1))
