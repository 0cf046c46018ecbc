#!/usr/bin/env python3
"""Diagnostic: per-kernel times of la_gpu_lz4_decode on an N-frame C2-shaped stream for a list of option
words (0 = default, 2 = no checksums, 8 = polling expand kernel, ...).
usage: python tools/exp_expand_time.py [frames] [opt,opt,...]      (LA_GPU_LIB picks another build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libarchive_amd as la
from libarchive_amd.lz4 import Lz4DevicePlan
import streams as S

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
opts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,2,8,10").split(",")]
if os.environ.get("LA_EXP_TEXT"):
    # text-like blocks through liblz4: thousands of short sequences per 64 KiB block (the segmented launch's work)
    import random
    rnd = random.Random(5)
    nw = int(os.environ["LA_EXP_TEXT"])
    words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(2, 9))) + b" " for _ in range(nw)]
    blocks = []
    for _ in range(64):
        d = bytearray()
        while len(d) < 65536:
            d += rnd.choice(words)
        d = bytes(d[:65536])
        blocks.append((d, S.lz4_block(S.lz4_compress_block(d), bsum=True)))
    one = b"".join(S.lz4_frame(blocks[i:i + 16], flg=0x74)[0] for i in range(0, 64, 16))
    img = np.frombuffer(one * (frames // 4), dtype=np.uint8).copy()
else:
    img, _ = S.synth_lz4_stream(0x4C413335, 0, frames, nthreads=16, want_plain=False)
idx = la.lz4_index(img)
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(img).cuda()
plan = Lz4DevicePlan(ctx, d_src, idx)
C, U = img.size, int(idx.max_out)
print("frames %d, blocks %d, C %.1f MB, U %.1f MB" % (frames, plan.n_blocks, C / 1e6, U / 1e6))
if os.environ.get("LA_EXP_TEXT"):
    ns = plan.nseq_host() if hasattr(plan, "nseq_host") else None
for o in opts:
    for _ in range(2):
        plan.run(o)
    ctx.sync()
    ctx.profile_enable(True)
    acc = {}
    K = 5
    for _ in range(K):
        plan.run(o); ctx.sync()
        for name, ms in ctx.profile_read():
            acc[name] = acc.get(name, 0.0) + ms / K
    ctx.profile_enable(False)
    e = acc.get("lz4_expand", 0.0)
    print("options %2d: %s | expand C+U %.0f GB/s = %.3f of 8 TB/s" % (o, " ".join("%s %.3f" % kv for kv in acc.items()), (C + U) / e / 1e6 if e else 0, (C + U) / e / 1e6 / 8000 if e else 0), flush=True)
