#!/usr/bin/env python3
"""A-level of the gzip filter: la_cat over a C3-shaped stream (BGZF-style 64 KiB members) in /dev/shm, process start and
PCIe included; with and without the second slab (LA_GZ_NO_COPY_AHEAD=1).  usage (GPU box): python tools/measure_gz_filter_path.py [decoded MiB]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrent.futures import ThreadPoolExecutor
import bench as B
import streams as S
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
uniq = 256
_, plain = S.synth_lz4_stream(B.SEED, 0, uniq, B.BPF, B.BLOCK, nthreads=16)
pieces = [(plain[i:i + B.BLOCK].tobytes(),) for i in range(0, plain.size, B.BLOCK)]
with ThreadPoolExecutor(16) as pool:
    members = list(pool.map(B._gz_make_member, pieces, chunksize=64))
one = b"".join(members)
path = "/dev/shm/la_measure.gz"
with open(path, "wb") as f:
    for _ in range(mib // uniq):
        f.write(one)
cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
for label, extra in (("two slabs (copy ahead)", {}), ("one slab (LA_GZ_NO_COPY_AHEAD=1)", {"LA_GZ_NO_COPY_AHEAD": "1"})):
    env = dict(os.environ, LA_GPU_BID="all", **extra)
    best = None
    for rep in range(3):
        t0 = time.time()
        r = subprocess.run([cat, "-b", "16777216", path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
        best = dt if best is None else min(best, dt)
    print("gzip la_cat %d MiB decoded (%d MiB compressed), %s: %.3f s -> %.0f MiB/s" % (mib, (len(one) * (mib // uniq)) >> 20, label, best, mib / best), flush=True)
os.unlink(path)
