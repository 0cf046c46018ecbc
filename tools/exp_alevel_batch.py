#!/usr/bin/env python3
"""Diagnostic: A-level time of la_cat against the filter's window size (LA_GPU_BATCH_MIB), 1 GiB and 16 GiB decoded, best of 3."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import streams as S
cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
path = "/dev/shm/la_measure.lz4"
for mib in (1024, 16384):
    img, _ = S.synth_lz4_stream(0x4C413335, 0, mib, 16, 65536, nthreads=16, want_plain=False)
    img.tofile(path)
    del img
    for batch in sys.argv[1:] or ("16", "32", "64", "128", "256"):
        env = dict(os.environ, LA_GPU_BID="all", LA_GPU_BATCH_MIB=batch)
        best = None
        for rep in range(3):
            t0 = time.time()
            r = subprocess.run([cat, "-b", "16777216", path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.time() - t0
            assert r.returncode == 0, r.stderr
            best = dt if best is None else min(best, dt)
        print("%6d MiB decoded, window %4s MiB: %.3f s -> %.0f MiB/s" % (mib, batch, best, mib / best), flush=True)
os.unlink(path)
