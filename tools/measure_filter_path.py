#!/usr/bin/env python3
"""A-level measurement: the whole drop-in path as bsdcat sees it -- file -> read core -> lz4/gzip
filter (H2D, device decode, D2H) -> archive_read_data_block -> /dev/null -- PCIe included.
This is NOT bench.py's `value` (that one starts with inputs resident in HBM); DESIGN.md quotes it.
usage (GPU box): python tools/measure_filter_path.py [decoded MiB]"""
import os, subprocess, sys, time, zlib, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import streams as S

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
frames = mib  # 16 blocks x 64 KiB = 1 MiB per frame
img, plain = S.synth_lz4_stream(0x4C413335, 0, frames, 16, 65536, nthreads=16, want_plain=False)
path = "/dev/shm/la_measure.lz4"
img.tofile(path)
cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
env = dict(os.environ)
for batch in ("32", "64", "128"):
    env["LA_GPU_BATCH_MIB"] = batch
    best = None
    for rep in range(3):
        t0 = time.time()
        r = subprocess.run([cat, path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
        best = dt if best is None else min(best, dt)
    print("lz4  la_cat %d MiB decoded, window %s MiB: %.3f s  -> %.0f MiB/s decoded (process start, file read, PCIe both ways included)"
          % (mib, batch, best, mib / best), flush=True)
os.unlink(path)
