import os, sys, time, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import zstd_support as Z
import libarchive_amd as la
from libarchive_amd import zstd as LZ
z = Z.libzstd(); rnd = random.Random(2)
d = b"".join(Z.gen(rnd, 65536, 2 if i % 2 else 4) for i in range(64)) * 4      # 16 MiB
img = Z.zstd_compress(z, d, 3)
ctx = la.GpuContext(0)
frames, end_kind, consumed, dst_bytes = LZ.index_image(img)
d_src = torch.from_numpy(np.frombuffer(img, dtype=np.uint8).copy()).cuda()
plan = LZ.ZstdDevicePlan(ctx, d_src, frames, dst_bytes)
plan.run(); res = plan.results(); assert (res["status"] == 0).all() and int(res["out_len"].sum()) == len(d)
assert plan.d_dst[:len(d)].cpu().numpy().tobytes() == d
ctx.sync(); t0 = time.time(); plan.run(); ctx.sync(); dt = time.time() - t0
print("ONE zstd frame of %d MiB decoded (level 3, %d blocks of 128 KiB): %.1f ms on one wave = %.1f MiB/s" % (len(d) >> 20, len(d) // 131072, dt * 1e3, len(d) / dt / 2**20))
