#!/usr/bin/env python3
"""Diagnostic: the gzip filter's own per-window trace (LA_GPU_TRACE=1) over a C3-shaped stream through la_cat, summed by phase."""
import os, re, subprocess, sys, time, collections
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
env = dict(os.environ, LA_GPU_BID="all", LA_GPU_TRACE="1")
t0 = time.time()
r = subprocess.run([cat, "-b", "16777216", path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
dt = time.time() - t0
os.unlink(path)
acc = collections.Counter(); n = 0
for line in r.stderr.decode().splitlines():
    m = re.search(r"gather ([\d.]+) ms, index \+ launch ([\d.]+) ms", line)
    if m: acc["gather"] += float(m.group(1)); acc["index + launch"] += float(m.group(2)); n += 1
    m = re.search(r"h2d\+decode ([\d.]+) ms, walk\+grow ([\d.]+) ms, d2h ([\d.]+) ms", line)
    if m: acc["wait for results (h2d + decode [+ copy ahead])"] += float(m.group(1)); acc["walk + grow"] += float(m.group(2)); acc["d2h (when not copied ahead)"] += float(m.group(3))
print("la_cat %d MiB decoded: %.3f s wall, %d windows; host thread inside the filter, summed over the windows:" % (mib, dt, n))
for k, v in acc.items():
    print("  %-50s %8.1f ms" % (k, v))
print("  %-50s %8.1f ms" % ("all of the above", sum(acc.values())))
