#!/usr/bin/env python3
"""Diagnostic: where the in-order LDS-window expand kernel (la_lz4_inorder.hip) spends its cycles, from
in-kernel counters of the LA_DIAG build (`make -C libarchive_amd/csrc diag`).  Not a benchmark: the
stamps cost time; read shares and per-group figures only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libarchive_amd._native as N
N.GPU_LIB_PATH = os.environ.get("LA_DIAG_LIB") or os.path.join(ROOT, "libarchive_amd", "csrc", "libla_gpu_diag.so")
import libarchive_amd as la
from libarchive_amd.lz4 import Lz4DevicePlan
import streams as S

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bpw = int(os.environ.get("IO_BPW", "8"))
img, _ = S.synth_lz4_stream(0x4C413335, 0, frames, nthreads=16, want_plain=False)
idx = la.lz4_index(img)
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(img).cuda()
plan = Lz4DevicePlan(ctx, d_src, idx)
nb = plan.n_blocks
nwg = (nb + bpw - 1) // bpw
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
plan.run(); ctx.sync()
assert la.gpu_lib().la_diag_set_io_stamps(C.c_void_p(stamps.data_ptr())) == 0
plan.run(); ctx.sync()
st = stamps.cpu().numpy().reshape(nwg, 16).astype(np.float64)
blocks = nb
tot = st.sum(axis=0)
groups = tot[3]
print("blocks %d, workgroups %d, groups of 64 sequences per block %.1f" % (blocks, nwg, groups / blocks))
print("M: cycles per block %.0f = per group %.0f; waiting for L %.0f per group; passes per group %.2f" %
      (tot[0] / blocks, tot[0] / groups, tot[1] / groups, tot[2] / groups))
print("M: pass 1 %.0f cycles per group, later rounds %.0f per group" % (tot[6] / groups, tot[7] / groups))
print("M: the rest (ring read, unpack, publish) %.0f per group" % ((tot[0] - tot[1] - tot[6] - tot[7]) / groups))
lw = int(os.environ.get("IO_LWAVES", "6"))
print("L (%d waves): %.0f cycles per block and wave in the group loop (%.0f per group of the wave), waiting for the window %.0f per block and wave" %
      (lw, tot[8] / blocks / lw, tot[8] / groups, tot[5] / blocks / lw))
print("L: waiting for a ring slot (for M) %.0f cycles per group of the wave; F: %.1f polls per block" % (tot[11] / groups, tot[9] / blocks))
t0, t1 = st[:, 12], st[:, 13]
ok = t0 > 0
span = (t1[ok].max() - t0[ok].min())
print("workgroups in flight on average: %.1f (sum of lifetimes %.0f / span %.0f ticks of 10 ns); mean lifetime %.1f us, span %.1f us" %
      ((t1[ok] - t0[ok]).sum() / span, (t1[ok] - t0[ok]).sum(), span, (t1[ok] - t0[ok]).mean() / 100, span / 100))
if tot[4] and tot[14]:
    print("LDS round trip seen by M (drained before and after): %.0f cycles at the head of pass 1, %.0f at the head of a later round" % (tot[14] / tot[4], tot[15] / max(tot[10], 1)))
print("M: wait for the first records of a block %.0f cycles per block (of all waiting for L: %.0f per block); F: tail after the last group %.0f cycles per block" % (tot[14] / blocks, tot[1] / blocks, tot[15] / blocks))
print("L: from the window being free to the block's first records %.0f cycles per block" % (tot[10] / blocks))
print("L, first iteration of a block: loads issued after %.0f cycles, literals stored after %.0f, records out after %.0f" % (tot[4] / blocks, tot[9] / blocks, tot[10] / blocks))
