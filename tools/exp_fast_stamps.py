#!/usr/bin/env python3
"""Diagnostic: phase shares of the LDS-window expand kernel from in-kernel stamps.
Needs `make -C libarchive_amd/csrc diag`.  Not a benchmark: read the SHARES only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libarchive_amd._native as N
N.GPU_LIB_PATH = os.environ.get("LA_DIAG_LIB") or os.path.join(ROOT, "libarchive_amd", "csrc", "libla_gpu_diag.so")
import libarchive_amd as la
from libarchive_amd.lz4 import Lz4DevicePlan
import streams as S

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512
img, _ = S.synth_lz4_stream(0x4C413335, 0, frames, nthreads=16, want_plain=False)
idx = la.lz4_index(img)
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(img).cuda()
plan = Lz4DevicePlan(ctx, d_src, idx)
nb = plan.n_blocks
stamps = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
plan.run(); ctx.sync()
assert la.gpu_lib().la_diag_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
plan.run(); ctx.sync()
st = stamps.cpu().numpy().reshape(nb, 8).astype(np.float64)
names = ["prepass+literals", "barrier1", "matches", "barrier2", "flush"]
tot = st[:, 5] - st[:, 0]
print("blocks %d  mean cycles per workgroup %.0f  (median %.0f)" % (nb, tot.mean(), np.median(tot)))
for i, nme in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print("  %-18s mean %8.0f  median %8.0f  share %.1f%%" % (nme, d.mean(), np.median(d), 100 * d.sum() / tot.sum()))
if os.environ.get("LA_DIAG_L"):
    # diag library built with -DLA_DIAG_L: stamps 6 / 7 are timestamps inside the first phase
    a = st[:, 6] - st[:, 0]; b = st[:, 7] - st[:, 6]; c = st[:, 1] - st[:, 7]
    print("  inside prepass+literals: entries loaded, chunk index and positions published, barrier %.0f; chunk's table entries back %.0f; literal stores %.0f"
          % (a.mean(), b.mean(), c.mean()))
    sys.exit(0)
if os.environ.get("LA_DIAG_M"):
    # diag library built with -DLA_DIAG_M: stamps 6 / 7 are timestamps inside the match phase
    a = st[:, 6] - st[:, 2]; b = st[:, 7] - st[:, 6]; c = st[:, 3] - st[:, 7]
    print("  inside matches: chunk map + walks + dependency ranges %.0f; last-sequence entries from global memory %.0f; the eight slots %.0f"
          % (a.mean(), b.mean(), c.mean()))
    sys.exit(0)
raw = stamps.cpu().numpy().reshape(nb, 8).astype(np.uint64)
M40 = np.uint64((1 << 40) - 1)
scan = (raw[:, 6] & M40).astype(np.float64); rounds = (raw[:, 6] >> np.uint64(40)).astype(np.float64)
copy = (raw[:, 7] & M40).astype(np.float64); bar = ((raw[:, 7] >> np.uint64(40)) << np.uint64(4)).astype(np.float64)
print("match phase per block (wave 0's view): rounds %.1f, look+push %.0f cycles (%.0f per round), queue work %.0f (%.0f per round), barrier waits %.0f (%.0f per round)"
      % (rounds.mean(), scan.mean(), scan.mean() / rounds.mean(), copy.mean(), copy.mean() / rounds.mean(), bar.mean(), bar.mean() / rounds.mean()))
