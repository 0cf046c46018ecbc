#!/usr/bin/env python3
"""zstd read path on one MI355X: device-resident decode rate of a many-frame stream (pzstd shape) through the C ABI,
the same stream through la_cat (file -> filter -> archive_read_data_block, PCIe and process start included), a one-frame
stream (this design's worst case), and the image's libzstd on one host core beside them.
usage: python tools/measure_zstd.py [frames=16384] [frame_kib=64] [level=3]"""
import ctypes, os, subprocess, sys, time
os.environ.setdefault("LA_GPU_BID", "all")   # the one-frame measurement is a shape the default bid policy declines
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import zstd_support as Z

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lvl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
z = Z.libzstd()
import random
rnd = random.Random(1)
# 64 distinct frames of text-like data, tiled
uniq = []
for i in range(64):
    d = Z.gen(rnd, kib * 1024, 2 if i % 2 else 4)
    uniq.append((d, Z.zstd_compress(z, d, lvl)))
img = b"".join(uniq[i % 64][1] for i in range(nfr))
plain_len = nfr * kib * 1024
print("stream: %d frames of %d KiB, level %d: %d compressed bytes, %d decoded (ratio %.2f)" % (nfr, kib, lvl, len(img), plain_len, plain_len / len(img)))

# host baseline first (before any GPU context): libzstd one-shot over a sample
buf = ctypes.create_string_buffer(64 * kib * 1024 + 64)
sample = b"".join(u[1] for u in uniq)
t0 = time.time(); reps = 0
while time.time() - t0 < 5:
    n = z.ZSTD_decompress(buf, len(buf), sample, len(sample)); reps += 1
cpu = reps * 64 * kib * 1024 / (time.time() - t0) / 2**20
print("libzstd %d, one core, same frames: %.0f MiB/s decoded" % (z.ZSTD_versionNumber(), cpu))

path = "/dev/shm/la_measure.zst"
open(path, "wb").write(img)
one = Z.zstd_compress(z, b"".join(u[0] for u in uniq[:16]), lvl)   # ONE frame of 16 x frame size
path1 = "/dev/shm/la_measure_one.zst"
open(path1, "wb").write(one)
cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
for p, what, nbytes in ((path, "la_cat, %d frames" % nfr, plain_len), (path1, "la_cat, ONE frame of %d KiB" % (16 * kib), 16 * kib * 1024)):
    best = None
    for _ in range(2):
        t0 = time.time()
        with open("/dev/null", "wb") as sink:      # (the bytes are compared in tests/test_gpu_zstd.py; here only the clock)
            out = subprocess.run([cat, p], stdout=sink, stderr=subprocess.PIPE)
        dt = time.time() - t0
        assert out.returncode == 0, (out.returncode, out.stderr[-200:])
        best = dt if best is None or dt < best else best
    print("%s: %.3f s = %.1f MiB/s decoded (process start + HIP init + PCIe both ways included)" % (what, best, nbytes / best / 2**20))
os.unlink(path); os.unlink(path1)

import torch
import libarchive_amd as la
from libarchive_amd import zstd as LZ
ctx = la.GpuContext(0)
frames, end_kind, consumed, dst_bytes = LZ.index_image(img)
assert len(frames) == nfr and consumed == len(img)
d_src = torch.from_numpy(np.frombuffer(img, dtype=np.uint8).copy()).cuda()
plan = LZ.ZstdDevicePlan(ctx, d_src, frames, dst_bytes)
plan.run(); res = plan.results()
assert (res["status"] == 0).all() and int(res["out_len"].sum()) == plain_len
got = plan.d_dst[:kib * 1024].cpu().numpy().tobytes()
assert got == uniq[0][0]
for _ in range(2):
    plan.run()
ctx.sync()
t0 = time.time(); K = 5
for _ in range(K):
    plan.run()
ctx.sync()
dt = (time.time() - t0) / K
print("la_gpu_zstd_decode, inputs resident in HBM: %.2f ms per pass = %.0f MiB/s decoded (%.1fx one host core)" % (dt * 1e3, plain_len / dt / 2**20, plain_len / dt / 2**20 / cpu))
