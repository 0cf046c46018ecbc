#!/usr/bin/env python3
"""BASELINE.json configs[3] shape, measured end to end as bsdtar -t sees it: a tar of N x 256 KiB entries behind
64 KiB gzip members (and the same tar behind lz4 frames) -> file -> read core -> device filter -> ustar walker ->
archive_read_next_header / archive_read_data_block, every body byte touched.  PCIe both ways and process start
included; NOT bench.py's `value`.  usage (GPU box): python tools/measure_tar_gz.py [entries] [window MiB ...]"""
import io, os, subprocess, sys, tarfile, time, zlib
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

ENTRY = 262144


def _body(i):
    rs = np.random.RandomState(i)
    words = rs.randint(0, 256, size=(4096, 8), dtype=np.uint8)
    return words[rs.randint(0, 4096, size=ENTRY // 8)].tobytes()


def _member(chunk):
    import streams as S
    return S.gz_member(chunk, level=1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    windows = sys.argv[2:] or ["64", "256"]
    bio = io.BytesIO()
    with tarfile.open(fileobj=bio, mode="w", format=tarfile.USTAR_FORMAT) as t:
        for i in range(n):
            ti = tarfile.TarInfo("set/%05d/entry%07d.bin" % (i // 1000, i))
            ti.size = ENTRY
            ti.mtime = 1700000000 + i
            t.addfile(ti, io.BytesIO(_body(i % 64)))
    tar = bio.getvalue()
    chunks = [tar[o:o + 65536] for o in range(0, len(tar), 65536)]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        gz = b"".join(ex.map(_member, chunks, chunksize=64))
    path = "/dev/shm/la_measure.tar.gz"
    open(path, "wb").write(gz)
    walk = os.path.join(ROOT, "libarchive_amd", "host", "la_tarwalk")
    print("tar %d entries x %d KiB = %.0f MiB decoded, %d gzip members, %.0f MiB compressed"
          % (n, ENTRY // 1024, len(tar) / 2**20, len(chunks), len(gz) / 2**20), flush=True)
    for w in windows:
        env = dict(os.environ, LA_GPU_BATCH_MIB=w)
        best = None
        for rep in range(3):
            t0 = time.time()
            r = subprocess.run([walk, "-b", str(4 << 20), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            dt = time.time() - t0
            assert r.returncode == 0, r.stderr
            best = dt if best is None else min(best, dt)
        line = r.stdout.decode().strip()
        assert ("%d entries, %d body bytes" % (n, n * ENTRY)) in line, line
        print("tar.gz la_tarwalk window %s MiB: %.3f s -> %.0f MiB/s decoded, %.0f entries/s   [%s]"
              % (w, best, len(tar) / 2**20 / best, n / best, line.split(": ", 1)[1]), flush=True)
    if os.environ.get("LA_MEASURE_TRACE"):
        env = dict(os.environ, LA_GPU_BATCH_MIB=windows[0], LA_GPU_TRACE="1")
        r = subprocess.run([walk, "-b", str(4 << 20), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        print(r.stderr.decode(), flush=True)
    os.unlink(path)


if __name__ == "__main__":
    main()
