#!/usr/bin/env python3
"""Randomized parity sweep of the whole FILTER path on the GPU box (not part of the pytest suite): 1 MiB gather windows
so that every stream crosses several windows (two in flight for lz4), frames / members / dependent-block frames
crossing the borders, mutations, truncations, junk; bytes, return code and error string against the oracle.
usage: python tools/fuzz_filters_gpu.py [seconds per codec]"""
import os, random, sys, time
os.environ.setdefault("LA_GPU_BID", "all")   # every stream shape is decoded here, lone units included (the bid policy has its own test)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LA_GPU_BATCH_MIB"] = "1"
import la_api
import oracle_lib as O
import streams as S
from test_gpu_filters import _dependent_frame

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0


def mutate(rnd, m):
    how = rnd.randrange(4)
    m = bytearray(m)
    if how == 1:
        m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
    elif how == 2:
        m = m[:rnd.randrange(1, len(m))]
    elif how == 3:
        m += bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 9)))
    return bytes(m), how


t0 = time.time(); n = 0; t = 0
while time.time() - t0 < budget:
    t += 1
    rnd = random.Random(770000 + t)
    if t % 4 == 0:
        words = [rnd.randbytes(rnd.randint(2, 11)) for _ in range(rnd.choice([30, 3000]))]
        data = b"".join(rnd.choice(words) for _ in range(rnd.randint(1000, 500000)))
        img, _ = _dependent_frame(data, flg=rnd.choice([0x44, 0x54, 0x40]))
        tail, _ = S.synth_lz4_stream(t, 0, 2, blocks_per_frame=2, block_size=3000, nthreads=1)
        img = img + tail.tobytes()
    else:
        img, _ = S.synth_lz4_stream(5000 + t, 0, rnd.randint(3, 14), blocks_per_frame=rnd.choice([1, 4, 7]),
                                    block_size=rnd.choice([4096, 30000, 65536]), nthreads=2)
        img = img.tobytes()
    m, how = mutate(rnd, img)
    out, res = O.lz4_stream_decode(m, 1 << 27)
    want = (out.tobytes(), res.rc, res.errmsg.decode())
    for rs in (None, 65536, 1000):
        got = la_api.as_reference_tuple(la_api.cat(m, read_size=rs))
        assert got == want, ("lz4", t, how, rs, len(got[0]), len(want[0]), got[1:], want[1:])
        n += 1
print("lz4 filter path: %d reads of %d streams agree with the oracle" % (n, t), flush=True)

t0 = time.time(); n = 0; t = 0
while time.time() - t0 < budget:
    t += 1
    rnd = random.Random(880000 + t)
    words = [rnd.randbytes(rnd.randint(1, 10)) for _ in range(200)]
    parts = []
    for k in range(rnd.randint(2, 30)):
        sz = rnd.choice([0, 1, 500, 20000, 65536, 70000, rnd.randint(0, 300000)])
        kind = rnd.randrange(3)
        d = (b"".join(rnd.choice(words) for _ in range(sz // 5 + 1))[:sz] if kind == 0 else
             rnd.randbytes(sz) if kind == 1 else bytes([k]) * sz)
        mem = bytearray(S.gz_member(d, level=rnd.choice([0, 1, 6, 9]), name=b"n%d" % k if rnd.random() < 0.3 else None))
        if rnd.random() < 0.1:
            mem[8], mem[9] = rnd.randrange(256), rnd.randrange(256)     # XFL / OS no writer emits
        parts.append(bytes(mem))
    m = bytearray(b"".join(parts))
    how = rnd.randrange(5)
    if how == 1:
        m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
    elif how == 2:
        m = m[:rnd.randrange(1, len(m))]
    elif how == 3:
        m += rnd.randbytes(rnd.randint(1, 20))
    elif how == 4 and len(parts) > 2:
        m[len(parts[0]) + len(parts[1]) // 2] ^= 0x04
    m = bytes(m)
    out, res = O.gzip_stream_decode(m, 1 << 27)
    want = (out.tobytes(), res.rc, res.errmsg.decode())
    for rs in (None, 4096):
        got = la_api.as_reference_tuple(la_api.cat(m, read_size=rs))
        assert got == want, ("gzip", t, how, rs, len(got[0]), len(want[0]), got[1:], want[1:])
        n += 1
print("gzip filter path: %d reads of %d streams agree with the oracle" % (n, t), flush=True)
