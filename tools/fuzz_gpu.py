#!/usr/bin/env python3
"""Longer randomized parity sweep on the GPU box (not part of the pytest suite): mutated and
truncated lz4 streams and deflate members, every kernel variant against the oracle.
usage: python tools/fuzz_gpu.py [seconds per codec]"""
import os, random, sys, time, zlib
os.environ.setdefault("LA_GPU_BID", "all")   # every stream shape is decoded here, lone units included (the bid policy has its own test)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import libarchive_amd as la
import oracle_lib as O
import streams as S
import test_gpu_lz4 as TL
import test_gpu_gzip as TG

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = la.GpuContext(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)

# ---- lz4: frames of real-compressor blocks (fast-path eligible), mutations in payloads / headers ----
t0 = time.time(); n = 0; seed = 1000
while time.time() - t0 < budget:
    seed += 1
    rnd = random.Random(seed)
    words = [rnd.randbytes(rnd.randint(1, 9)) for _ in range(rnd.choice([3, 20, 200]))]
    blocks = []
    for k in range(rnd.randint(1, 8)):
        nbytes = rnd.choice([1, 17, 300, 5000, 65536, rnd.randint(1, 65536)])
        kind = rnd.randrange(4)
        d = (b"".join(rnd.choice(words) for _ in range(nbytes // 3 + 1))[:nbytes] if kind == 0 else
             rnd.randbytes(nbytes) if kind == 1 else bytes([k]) * nbytes if kind == 2 else (b"abc" * nbytes)[:nbytes])
        c = bytearray(S.lz4_compress_block(d))
        if rnd.random() < 0.5:
            for _ in range(rnd.randint(1, 3)):
                c[rnd.randrange(len(c))] = rnd.getrandbits(8)
        blocks.append((d, S.lz4_block(bytes(c), bsum=rnd.random() < 0.3)))
    img, _ = S.lz4_frame(blocks, flg=rnd.choice([0x60, 0x64, 0x70, 0x74]))
    img = bytearray(img)
    if rnd.random() < 0.2:
        img[rnd.randrange(len(img))] ^= 1 << rnd.randrange(8)
    if rnd.random() < 0.2:
        img = img[:rnd.randrange(1, len(img))]
    img = bytes(img)
    ref, res = O.lz4_stream_decode(img, 1 << 22)
    got = TL.gpu_decode(ctx, img)
    assert got == (ref.tobytes(), res.rc, res.errmsg.decode()), ("lz4", seed)
    n += 1
print("lz4: %d randomized streams, all kernel variants == oracle" % n, flush=True)

# ---- deflate: members with slots of at most 64 KiB (two-phase eligible), mutated / truncated ----
t0 = time.time(); n = 0; seed = 5000
while time.time() - t0 < budget:
    seed += 1
    rnd = random.Random(seed)
    words = [rnd.randbytes(rnd.randint(1, 12)) for _ in range(rnd.choice([4, 40, 400]))]
    bodies, caps, expect = [], [], []
    for t in range(200):
        nbytes = rnd.choice([0, 1, 100, 3000, 65536, rnd.randint(0, 65536)])
        kind = rnd.randrange(4)
        d = (b"".join(rnd.choice(words) for _ in range(nbytes // 5 + 1))[:nbytes] if kind == 0 else
             rnd.randbytes(nbytes) if kind == 1 else bytes([t & 255]) * nbytes if kind == 2 else (b"ab" * nbytes)[:nbytes])
        c = bytearray(TG.deflate(d, rnd.choice([0, 1, 6, 9]),
                                 rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])))
        if rnd.random() < 0.6:
            for _ in range(rnd.randint(1, 3)):
                if c:
                    c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
        if rnd.random() < 0.2 and len(c) > 1:
            c = c[:rnd.randrange(1, len(c))]
        cap = rnd.choice([65536, 65536, len(d), max(len(d) - 1, 0), rnd.randint(0, 65536)])
        c = bytes(c)
        rc, cons, out = O.inflate_raw(c, cap)
        bodies.append(c); caps.append(cap); expect.append((rc, cons, out))
    try:
        res, _ = TG.gpu_inflate(ctx, bodies, caps, verify=False)
    except AssertionError:
        rs = {opt: TG._gpu_inflate(ctx, bodies, caps, False, opt)[0] for opt in (2, 4, 8)}
        for i in range(len(bodies)):
            v = {opt: (rs[opt][i][0], len(rs[opt][i][1]), zlib.crc32(rs[opt][i][1])) for opt in rs}
            if len(set(v.values())) > 1:
                print("MISMATCH seed", seed, "member", i, "cap", caps[i], "len(body)", len(bodies[i]), "oracle", expect[i][0], len(expect[i][2]), "paths", v, flush=True)
        raise
    for i, ((st, out, cons, crc), (rc, ocons, oout)) in enumerate(zip(res, expect)):
        want = {0: (TG.ST_OK, TG.ST_NOTRAILER), 1: (TG.ST_TRUNC,), 2: (TG.ST_DATA,), 3: (TG.ST_FULL,)}[rc]
        assert st in want, ("gz status", seed, i, st, rc)
        if rc != 3:
            assert out == oout, ("gz bytes", seed, i, st, rc, len(out), len(oout))
        if rc == 0:
            assert cons == ocons, ("gz consumed", seed, i)
    n += len(bodies)
print("deflate: %d randomized members, three kernel paths agree and == oracle" % n, flush=True)
