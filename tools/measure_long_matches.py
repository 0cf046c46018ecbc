import sys, time; sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np, torch
import libarchive_amd as la
from libarchive_amd import _native as N
from libarchive_amd.lz4 import Lz4DevicePlan
import streams as S
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for name, blk in (("zeros", bytes(65536)), ("period7", (b"abcdefg" * 9400)[:65536]), ("runs", b"".join(bytes([i & 255]) * 997 for i in range(66))[:65536])):
    one = S.lz4_block(S.lz4_compress_block(blk), bsum=True)
    img, plain = S.lz4_frame([(blk, one)] * 16, flg=0x74)
    img = np.frombuffer(img * 256, dtype=np.uint8)     # 4096 blocks = 256 MiB decoded
    idx = la.lz4_index(img)
    plan = Lz4DevicePlan(ctx, torch.from_numpy(img.copy()).cuda(), idx)
    for opt, tag in ((0, "default"), (N.LA_LZ4_OPT_EXPAND_INORDER, "in-order"), (N.LA_LZ4_OPT_GENERAL_ONLY, "general")):
        plan.run(opt); ctx.sync()
        t0 = time.time(); plan.run(opt); ctx.sync(); dt = time.time() - t0
        sm = plan.summary()
        ok = int(sm["n_bad_units"]) == 0 and plan.d_dst[:65536].cpu().numpy().tobytes() == blk
        print("%-8s %-8s %8.2f ms  (%d blocks, ok %s) -> %.1f GB/s" % (name, tag, dt * 1e3, len(idx.blocks), ok, 65536 * len(idx.blocks) / dt / 1e9), flush=True)
