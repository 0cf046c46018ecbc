#!/usr/bin/env python3
"""Randomized zstd streams through the product's filter path (la_api.cat -> la_filter_zstd.c -> la_gpu_zstd_decode), both
device kernels, against the oracle (oracle/orc_zstd.c, checker only): clean, bit-damaged and truncated multi-frame
streams made with the image's libzstd.  Outside the pytest suite; the result is committed under profiles/.
usage (GPU box): python tools/fuzz_zstd_gpu.py [seconds=60]"""
import os, random, sys, time
os.environ.setdefault("LA_GPU_BID", "all")   # every stream shape is decoded here, lone units included (the bid policy has its own test)
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import la_api
import zstd_support as Z

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
z, o = Z.libzstd(), Z.oracle_lib()
rnd = random.Random(0xF00D)
t0 = time.time()
n = clean = refused = strict_only = 0
while time.time() - t0 < budget:
    nf = rnd.randint(1, 4)
    plains = [Z.gen(rnd, rnd.choice([0, 3, 200, 5000, 70000, 140000, 300000]), rnd.randint(0, 4)) for _ in range(nf)]
    parts = []
    for i, d in enumerate(plains):
        parts.append(Z.zstd_compress(z, d, rnd.choice([-5, 1, 3, 6, 12, 19])))
        if rnd.random() < 0.2:
            parts.append(Z.skippable(bytes(rnd.randrange(30)), rnd.randrange(16)))
    img = bytearray(b"".join(parts))
    mode = rnd.random()
    if mode < 0.3 and len(img) > 2:
        img = img[:rnd.randrange(1, len(img))]
    elif mode < 0.8:
        for _ in range(rnd.randint(1, 3)):
            img[rnd.randrange(len(img))] ^= 1 << rnd.randrange(8)
    img = bytes(img)
    rc, out, msg = Z.oracle_decode(o, img, sum(map(len, plains)) + 600000)
    for lane_kernel in ("0", "1"):
        os.environ["LA_ZSTD_LANE_KERNEL"] = lane_kernel
        res = la_api.cat(img) if la_api._lib().archive_read_new else None
        data, grc, gmsg = la_api.as_reference_tuple(res)
        if res.open_rc != 0 or not res.filters or res.filters[0][1] != "zstd":
            # the damage hit the magic number: no zstd bidder, the bytes pass through raw (the reference does the same)
            continue
        if rc == 0:
            assert (data, grc) == (out, 0), ("accepted stream differs", n, lane_kernel, grc, gmsg)
        else:
            assert grc == la_api.ARCHIVE_FATAL, ("oracle refuses, product accepts", n, lane_kernel, msg)
            assert out.startswith(data), ("bytes in front of the error are not a prefix of the oracle's", n, lane_kernel)
            if msg == "Truncated zstd input":
                assert gmsg == msg, (n, gmsg)
            else:
                assert gmsg.startswith("Zstd decompression failed: ") or gmsg.startswith("zstd frame too large"), (n, gmsg)
    n += 1
    clean += rc == 0
    refused += rc != 0
print("zstd: %d randomized streams (%d accepted, %d refused), wave-per-frame and lane-per-frame kernels == oracle (bytes, verdict, message class)" % (n, clean, refused))
