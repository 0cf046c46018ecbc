#!/usr/bin/env python3
"""Diagnostic: where the wave-per-member inflate kernel spends its cycles (in-kernel cycle counters per member).
Needs `make -C libarchive_amd/csrc diag`.  Not a benchmark: the counters themselves cost time; read the SHARES."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libarchive_amd._native as N
N.GPU_LIB_PATH = os.path.join(ROOT, "libarchive_amd", "csrc", "libla_gpu_diag.so")
import libarchive_amd as la
from libarchive_amd.gzip import GzDevicePlan
import streams as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_, plain = S.synth_lz4_stream(0x4C413335, 0, n // 16, 16, 65536, nthreads=16)
img = np.frombuffer(b"".join(S.gz_member(plain[i:i + 65536].tobytes(), level=6) for i in range(0, plain.size, 65536)), dtype=np.uint8)
idx = la.gz_index(img, at_eof=True)
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(img.copy()).cuda()
plan = GzDevicePlan(ctx, d_src, idx)
nm = len(idx.members)
stamps = torch.zeros(nm * 8, dtype=torch.int64, device="cuda")
plan.run(N.LA_GZ_OPT_WAVE_KERNEL if hasattr(N, "LA_GZ_OPT_WAVE_KERNEL") else 2); ctx.sync()
assert la.gpu_lib().la_diag_set_inflate_stamps(C.c_void_p(stamps.data_ptr())) == 0
plan.run(2); ctx.sync()
st = stamps.cpu().numpy().reshape(nm, 8).astype(np.float64)
tot, sym, mat = st[:, 4], st[:, 6], st[:, 7]
print("members %d, 64 KiB each: %.0f symbols (%.0f matches) per member, %.0f cycles per member = %.0f per symbol"
      % (nm, sym.mean(), mat.mean(), tot.mean(), tot.mean() / sym.mean()))
for k, name in enumerate(["literal/length decode (incl. refill)", "literal bookkeeping", "length + distance decode", "flush + match copy"]):
    per = st[:, k].mean() / (sym.mean() if k < 2 else mat.mean())
    print("  %-38s %5.1f %% of the member, %6.0f cycles per %s" % (name, 100 * st[:, k].sum() / tot.sum(), per, "symbol" if k < 2 else "match"))
print("  %-38s %5.1f %%" % ("block headers, table builds, rest", 100 * (1 - st[:, :4].sum() / tot.sum())))
