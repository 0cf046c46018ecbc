#!/usr/bin/env python3
"""A-level measurement of the ZIP reader: a .zip of N deflate entries of 256 KiB (C4's entry shape)
through la_tarwalk -- file -> read core -> ZIP reader (central directory on the host, batched raw inflate + CRC32 on
the device) -> archive_read_next_header / archive_read_data_block -- process start, HIP init and PCIe included.
usage (GPU box): python tools/measure_zip.py [entries] [KiB per entry]"""
import os, subprocess, sys, time, zipfile, io, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import streams as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
_, plain = S.synth_lz4_stream(0x4C413335, 0, max(1, n * kib // 1024), 16, 65536, nthreads=16)
plain = plain.tobytes()
path = "/dev/shm/la_measure.zip"
t0 = time.time()
from concurrent.futures import ThreadPoolExecutor
import zlib, struct
def comp(i):
    d = plain[i * kib * 1024:(i + 1) * kib * 1024]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    return c.compress(d) + c.flush(), zlib.crc32(d), len(d)
with ThreadPoolExecutor(16) as ex:
    parts = list(ex.map(comp, range(n)))
# write the archive by hand (zipfile would compress again, single-threaded)
out = io.BytesIO(); cd = []
for i, (body, crc, usz) in enumerate(parts):
    name = ("dir/entry%06d.bin" % i).encode()
    off = out.tell()
    out.write(struct.pack("<IHHHHHIIIHH", 0x04034b50, 20, 0, 8, 0, 0x5821, crc, len(body), usz, len(name), 0) + name + body)
    cd.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014b50, 0x031e, 20, 0, 8, 0, 0x5821, crc, len(body), usz, len(name), 0, 0, 0, 0, 0o100644 << 16, off) + name)
cdo = out.tell(); out.write(b"".join(cd)); cds = out.tell() - cdo
assert n < 65535
out.write(struct.pack("<IHHHHIIH", 0x06054b50, 0, 0, n, n, cds, cdo, 0))
open(path, "wb").write(out.getvalue())
zipfile.ZipFile(path).testzip()
print("zip: %d entries x %d KiB = %d MiB decoded, %d MiB archive (built in %.1f s)" % (n, kib, n * kib // 1024, os.path.getsize(path) >> 20, time.time() - t0), flush=True)
walk = os.path.join(ROOT, "libarchive_amd", "host", "la_tarwalk")
best = None
for rep in range(3):
    t0 = time.time()
    r = subprocess.run([walk, "-b", str(4 << 20), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr
    best = dt if best is None else min(best, dt)
line = r.stdout.decode().strip()
assert ("%d entries, %d body bytes" % (n, n * kib * 1024)) in line, line
print("zip la_tarwalk: %.3f s -> %.0f MiB/s decoded, %.0f entries/s   [%s]" % (best, n * kib / 1024 / best, n / best, line.split(": ", 1)[1]), flush=True)
os.unlink(path)
