#!/usr/bin/env python3
"""Diagnostic: ONE lz4 frame of many 64 KiB blocks through la_cat (bounded windows), with the
content checksum that serialises its hashing.  usage: python tools/measure_single_frame.py [MiB]"""
import os, subprocess, sys, time
os.environ.setdefault("LA_GPU_BID", "all")   # these measure the lone-unit shapes the default bid policy declines
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import streams as S
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img, _ = S.synth_lz4_stream(0x4C413335, 0, 1, mib * 16, 65536, nthreads=16, want_plain=False)
path = "/dev/shm/la_single.lz4"
img.tofile(path)
cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
env = dict(os.environ, LA_GPU_BATCH_MIB="64")
best = None
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([cat, path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr
    best = dt if best is None else min(best, dt)
print("single frame, %d MiB decoded in %d blocks, content checksum on, window 64 MiB: %.3f s -> %.0f MiB/s" % (mib, mib * 16, best, mib / best))
os.unlink(path)
