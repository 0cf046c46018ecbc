#!/usr/bin/env python3
"""Diagnostic: frames of 4 MiB independent blocks (the lz4 CLI's default block size): they take
the general wave-per-block kernel.  Device-resident timing of la_gpu_lz4_decode.
usage: python tools/measure_big_blocks.py [decoded MiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import libarchive_amd as la
from libarchive_amd.lz4 import Lz4DevicePlan
import streams as S
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
import numpy as np
# plain data from the synthetic generator (64 KiB-block stream), re-cut into 4 MiB blocks and
# compressed by liblz4; frames of four blocks, BD = 7 (4 MiB), no content checksum (flg 0x60)
_, plain = S.synth_lz4_stream(0x4C413335, 0, mib, 16, 65536, nthreads=16)
plain = plain.tobytes()
parts = []
for f in range(0, len(plain), 16 << 20):
    blocks = []
    for b in range(f, min(f + (16 << 20), len(plain)), 4 << 20):
        d = plain[b:b + (4 << 20)]
        blocks.append((b"", S.lz4_block(S.lz4_compress_block(d))))
    fr, _ = S.lz4_frame(blocks, flg=0x60, bd=0x70)
    parts.append(fr)
img = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
frames = len(parts)
idx = la.lz4_index(img)
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(img).cuda()
plan = Lz4DevicePlan(ctx, d_src, idx)
plan.run(); ctx.sync()
ctx.profile_enable(True)
plan.run()
print('phases (ms):', {k: round(v, 2) for k, v in ctx.profile_read()})
ctx.profile_enable(False)
t0 = time.time()
for _ in range(3):
    plan.run()
ctx.sync()
dt = (time.time() - t0) / 3
sm = plan.summary()
print("%d frames x 4 blocks of 4 MiB, %d MiB decoded: %.1f ms per pass -> %.0f MiB/s (bad units %d, bad frames %d)"
      % (frames, int(sm["total_out"]) >> 20, dt * 1e3, (int(sm["total_out"]) >> 20) / dt, int(sm["n_bad_units"]), int(sm["n_bad_frames"])))
