#!/usr/bin/env python3
"""BASELINE.json configs[0] shape on the device path: ONE gzip member (no parallel axis: a deflate stream is a serial
bit chain, so one lane of one wave decodes it).  Reported so that the limit is on record, not as a result to be proud of.
usage (GPU box): python tools/measure_single_member_gz.py [MiB ...]"""
import os, subprocess, sys, time, zlib
os.environ.setdefault("LA_GPU_BID", "all")   # these measure the lone-unit shapes the default bid policy declines
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import streams as S

cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
for mib in [int(x) for x in sys.argv[1:]] or [8, 32]:
    rs = np.random.RandomState(mib)
    words = rs.randint(0, 256, size=(4096, 8), dtype=np.uint8)
    data = words[rs.randint(0, 4096, size=(mib << 20) // 8)].tobytes()
    t0 = time.time(); gz = S.gz_member(data, level=6); t_comp = time.time() - t0
    t0 = time.time(); assert zlib.decompress(gz, 31) == data; t_cpu = time.time() - t0
    path = "/dev/shm/la_single.gz"
    open(path, "wb").write(gz)
    t0 = time.time()
    r = subprocess.run([cat, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    dt = time.time() - t0
    assert r.returncode == 0 and r.stdout == data, r.stderr[-500:]
    print("single member %d MiB decoded (%d MiB compressed): la_cat %.2f s -> %.1f MiB/s; zlib on one host core %.2f s -> %.0f MiB/s"
          % (mib, len(gz) >> 20, dt, mib / dt, t_cpu, mib / t_cpu), flush=True)
    os.unlink(path)
