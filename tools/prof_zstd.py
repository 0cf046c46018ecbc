#!/usr/bin/env python3
"""The device-resident part of tools/measure_zstd.py alone (no child processes), for
`rocprofv3 --kernel-trace --stats -- python3 tools/prof_zstd.py`: 16 384 zstd frames of 64 KiB, 2 + 5 passes."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import zstd_support as Z
import libarchive_amd as la
from libarchive_amd import zstd as LZ
z = Z.libzstd(); rnd = random.Random(1)
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
uniq = [Z.zstd_compress(z, Z.gen(rnd, 65536, 2 if i % 2 else 4), 3) for i in range(64)]
img = b"".join(uniq[i % 64] for i in range(nfr))
ctx = la.GpuContext(0)
frames, end_kind, consumed, dst_bytes = LZ.index_image(img)
d_src = torch.from_numpy(np.frombuffer(img, dtype=np.uint8).copy()).cuda()
plan = LZ.ZstdDevicePlan(ctx, d_src, frames, dst_bytes)
for _ in range(2):
    plan.run()
res = plan.results()
assert (res["status"] == 0).all()
t0 = time.time()
for _ in range(5):
    plan.run()
ctx.sync()
print("%d frames: %.2f ms per pass" % (nfr, (time.time() - t0) / 5 * 1e3))
