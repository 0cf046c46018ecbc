#!/usr/bin/env python3
"""Diagnostic: fixed against per-byte cost of the A-level path (la_cat over a .lz4 in /dev/shm): several sizes, best of 3."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import streams as S
cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
env = dict(os.environ, LA_GPU_BID="all")
path = "/dev/shm/la_measure.lz4"
for mib in (64, 1024, 4096, 16384):
    img, _ = S.synth_lz4_stream(0x4C413335, 0, mib, 16, 65536, nthreads=16, want_plain=False)
    img.tofile(path)
    for args in ([], ["-b", "16777216"]):
        best = None
        for rep in range(3):
            t0 = time.time()
            r = subprocess.run([cat] + args + [path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.time() - t0
            assert r.returncode == 0, r.stderr
            best = dt if best is None else min(best, dt)
        print("%6d MiB decoded (%5d MiB compressed) %-14s %.3f s -> %.0f MiB/s" % (mib, img.size >> 20, " ".join(args) or "10k blocks", best, mib / best), flush=True)
    del img
os.unlink(path)
