#!/usr/bin/env python3
"""bench.py -- headline benchmark of the lz4 read-filter hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): raw-format + lz4 filter over a synthetic
many-block .lz4 stream -- concatenated frames of 16 independent 64 KiB blocks, block
checksums + content checksums + header check bytes all verified (XXH32) -- 16 GiB decoded
per GPU.  One "step" = one full pass of the device data plane over the whole resident
stream: la_gpu_lz4_decode() = block-checksum kernel, measure kernel, scan, expand kernel,
frame-checksum kernel, summary.  Inputs (compressed stream + host-built block table) are
resident in HBM before the timed region; outputs stay in HBM.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
With --gpus N > 1 and no WORLD_SIZE in the environment, bench.py starts the N ranks itself
(torch.distributed.run as a CHILD process, before anything here has touched a GPU).
N ranks = ONE stream of N x 16 GiB decoded (BASELINE configs[4] at N = 8), indexed once on the
host and cut on frame boundaries into N contiguous ranges balanced by C + U
(libarchive_amd/shard.py); every rank uploads and decodes only its range: weak scaling, no
data-path collective, ranks exchange only their batch summary.  --gather adds the explicit
gather of the decoded ranges to rank 0 over RCCL (the "single stream split" mode), timed
separately and reported beside the sharded number.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      -- dominant kernel (lz4 expand): algorithmic bytes C+U per launch divided by
                   its average duration measured with HIP events on the work stream
  cpu_baseline  -- the CPU oracle (a port of the reference filter) on a bounded sample,
                   1 thread, on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 0x4C413335
BLOCK = 65536
BPF = 16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gib", type=float, default=16.0, help="decoded GiB per GPU")
    ap.add_argument("--unique-mib", type=int, default=1024, help="unique decoded MiB generated on the host, tiled on device")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024, help="decoded MiB the CPU baseline decodes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the gzip C3 measurement the default lz4 line carries as `secondary`")
    ap.add_argument("--api-mib", type=int, default=16384, help="decoded MiB of the A-level (la_cat) measurement, 0 = skip")
    ap.add_argument("--general-only", action="store_true", help="force the general expand kernel")
    ap.add_argument("--extra-options", type=int, default=0, help="diagnostic: extra LA_LZ4_OPT_* bits")
    ap.add_argument("--gather", action="store_true", default=os.environ.get("LA_BENCH_GATHER", "") == "1",
                    help="N > 1: also time one gather of all decoded ranges to rank 0 (needs about N x 16 GiB free on GPU 0)")
    ap.add_argument("--workload", choices=["lz4", "gzip"], default="lz4",
                    help="lz4 = BASELINE configs[1] (the headline); gzip = configs[2] shape (64 KiB BGZF-style members)")
    return ap.parse_args()


def kernel_source_id():
    """Identifies the build of the device library a PMC traffic figure belongs to: the SHA-256 prefix of the
    libla_gpu.so in use (hipcc output is reproducible; comments do not change it, any kernel change does)."""
    import hashlib
    from libarchive_amd import _native as N
    return hashlib.sha256(open(N.GPU_LIB_PATH, "rb").read()).hexdigest()[:16]


def build_tiled_stream(torch, la, S, rank, world, unique_mib, total_gib, device):
    """Host: generate + index the unique region once; the stream is that region tiled
    (tiles x world times), its index the region's index tiled.  The splitter cuts the WHOLE
    stream's frame list; this rank materialises only its own byte range on its device."""
    from libarchive_amd import _native as N
    from libarchive_amd.shard import slice_index, split_stream
    frames_unique = max(1, (unique_mib << 20) // (BPF * BLOCK))
    tiles = max(1, int(round(total_gib * (1 << 30) / (frames_unique * BPF * BLOCK))))
    t0 = time.time()
    img, plain = S.synth_lz4_stream(SEED, 0, frames_unique, BPF, BLOCK,
                                    nthreads=min(16, os.cpu_count() or 1), want_plain=False)
    t_gen = time.time() - t0
    t0 = time.time()
    idx = la.lz4_index(img, at_eof=True)
    t_index = time.time() - t0
    assert idx.end_kind == N.LA_END_EOF and len(idx.frames) == frames_unique
    nb, nf, clen = len(idx.blocks), len(idx.frames), int(img.size)
    # the whole stream's tables (world x tiles copies of the region's)
    T = tiles * world
    blocks = np.tile(idx.blocks, T)
    frames = np.tile(idx.frames, T)
    t_of_b = np.repeat(np.arange(T, dtype=np.uint64), nb)
    t_of_f = np.repeat(np.arange(T, dtype=np.uint64), nf)
    blocks["src_off"] += t_of_b * np.uint64(clen)
    frames["desc_off"] += t_of_f * np.uint64(clen)
    frames["first_block"] += (t_of_f * np.uint64(nb)).astype(np.uint32)
    whole = N.Lz4Index(blocks, frames, idx.end_kind, clen * T, idx.max_out * T)
    f_lo, f_hi = split_stream(whole, clen * T, world)[rank]
    mine, b_lo, b_hi = slice_index(whole, clen * T, f_lo, f_hi)
    del blocks, frames, whole
    # this rank's bytes [b_lo, b_hi) of the stream, assembled on the device from the region's image
    d_unique = torch.from_numpy(img).to(device)
    parts, pos = [], b_lo
    while pos < b_hi:
        o = pos % clen
        n = min(clen - o, b_hi - pos)
        parts.append(d_unique[o:o + n])
        pos += n
    d_src = torch.cat(parts) if len(parts) != 1 else parts[0].clone()
    del parts, d_unique
    info = dict(frames_unique=frames_unique, tiles=tiles, gen_s=t_gen, index_s_unique=t_index,
                unique_compressed=clen, unique_blocks=nb, frame_range=(f_lo, f_hi), byte_range=(b_lo, b_hi),
                first_tile_frame=f_lo % nf)
    return img, idx, d_src, mine, info


def _timed_decodes(O, sample, cap, budget_s):
    reps, dt, t0, out = 0, 0.0, time.time(), None
    while dt < budget_s:
        out, res = O.lz4_stream_decode(sample, cap)
        assert res.rc == 0
        reps += 1
        dt = time.time() - t0
    return reps, dt, out


def cpu_baseline(S, O, img_unique, idx_unique, sample_mib, budget_s=8.0):
    """CPU side of the same run, on a bounded sample of the same stream:
    (i) the oracle = port of the reference lz4 filter incl. all XXH32 checks, 1 thread (libarchive is
    single-threaded per archive); (ii) the same framing and checks with the box's own liblz4
    LZ4_decompress_safe behind it (dlopen) -- what libarchive's filter executes; (iii) N independent
    replicas of (ii) (or (i) without liblz4) on all host cores, N stated."""
    import ctypes as C
    import threading
    nframes = len(idx_unique.frames)
    want = max(1, min(nframes, (sample_mib << 20) // (BPF * BLOCK)))
    end = int(idx_unique.frames["desc_off"][want] - 4) if want < nframes else int(img_unique.size)
    sample = img_unique[:end]
    cap = want * BPF * BLOCK + 64
    out, res = O.lz4_stream_decode(sample, cap)
    assert res.rc == 0 and len(out) == want * BPF * BLOCK
    mib = len(out) / (1 << 20)
    reps, dt, _ = _timed_decodes(O, sample, cap, budget_s)
    line = dict(value=round(reps * mib / dt, 1), unit="MiB/s", cores=1, kind="port",
                sample="%d x %d MiB decoded (%d frames of the same stream), oracle lz4 filter with block+content XXH32 checks, %.1f s of CPU work"
                       % (reps, int(mib), want, dt))
    variants = []
    have_liblz4 = False
    try:
        lz = C.CDLL("liblz4.so.1")
        O.lib().orc_set_external_lz4(C.cast(lz.LZ4_decompress_safe, C.c_void_p), C.cast(lz.LZ4_decompress_safe_usingDict, C.c_void_p))
        have_liblz4 = True
        out2, res2 = O.lz4_stream_decode(sample, cap)
        assert res2.rc == 0 and np.array_equal(out2, out)
        reps, dt, _ = _timed_decodes(O, sample, cap, budget_s * 0.6)
        variants.append(dict(value=round(reps * mib / dt, 1), unit="MiB/s", cores=1, kind="liblz4",
                             sample="same sample and framing, blocks decoded by the box's liblz4 %s LZ4_decompress_safe, XXH32 checks by the port, %.1f s"
                                    % (C.cast(lz.LZ4_versionString, C.CFUNCTYPE(C.c_char_p))().decode(), dt)))
    except (OSError, AttributeError) as e:
        variants.append(dict(kind="liblz4", error="not loadable on this box: %s" % e))
    # all host cores: N independent replicas (threads; the C call releases the GIL), same sample each
    ncores = os.cpu_count() or 1
    counts = [0] * ncores
    rwant = min(want, 64)          # 64 MiB per decode: every core finishes several inside the budget
    rend = int(idx_unique.frames["desc_off"][rwant] - 4) if rwant < nframes else int(img_unique.size)
    rsample, rcap, rmib = img_unique[:rend], rwant * BPF * BLOCK + 64, rwant * BPF * BLOCK / (1 << 20)
    stop = time.time() + budget_s * 0.75

    def replica(i):
        while time.time() < stop:
            o, r = O.lz4_stream_decode(rsample, rcap)
            assert r.rc == 0 and len(o) == rwant * BPF * BLOCK
            counts[i] += 1
    t0 = time.time()
    th = [threading.Thread(target=replica, args=(i,)) for i in range(ncores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    variants.append(dict(value=round(sum(counts) * rmib / dt, 1), unit="MiB/s", cores=ncores,
                         kind="liblz4 replicas" if have_liblz4 else "port replicas",
                         sample="%d independent replicas of the same decode on all %d host cores, %d decodes of %d MiB (the sample's first frames) in %.1f s"
                                % (ncores, ncores, sum(counts), int(rmib), dt)))
    if have_liblz4:
        O.lib().orc_set_external_lz4(None, None)
    line["variants"] = variants
    return line, out


def api_level(S, args):
    """A-level: the whole drop-in path as bsdcat drives it -- file in /dev/shm -> read core ->
    lz4 filter (gather, H2D, device decode, D2H) -> archive_read_data_block -> /dev/null, PCIe both ways,
    process start and HIP initialisation included.  Runs la_cat as a CHILD process before this process has
    touched the GPU.  Never `value`."""
    import subprocess
    mib = args.api_mib
    cat = os.path.join(ROOT, "libarchive_amd", "host", "la_cat")
    if mib <= 0 or not os.path.exists(cat) or not os.path.isdir("/dev/shm"):
        return None
    img, _ = S.synth_lz4_stream(SEED, 0, mib, BPF, BLOCK, nthreads=min(64, os.cpu_count() or 1), want_plain=False)
    path = "/dev/shm/la_bench_%d.lz4" % os.getpid()
    try:
        img.tofile(path)
        res = {}
        for name, argv in (("bsdcat_10k_blocks", [cat, path]), ("block_size_16m", [cat, "-b", str(16 << 20), path])):
            best = None
            for _ in range(2):
                t0 = time.time()
                r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
                dt = time.time() - t0
                if r.returncode != 0:
                    return dict(error=r.stderr.decode(errors="replace")[-300:])
                best = dt if best is None else min(best, dt)
            res[name] = round(mib / best, 1)
        return dict(unit="MiB/s decoded", program="la_cat (bsdcat shape) over a %d MiB-decoded .lz4 in /dev/shm, best of 2, process start + HIP init + PCIe both ways included" % mib,
                    compressed_bytes=int(img.size), **res)
    finally:
        if os.path.exists(path):
            os.unlink(path)


def _gz_make_member(args):
    import struct, zlib
    data, = args
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9)
    body = co.compress(data) + co.flush()
    total = 18 + len(body) + 8
    hdr = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\x03" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1)
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))



def measured_copy_peak(ctx, torch, mib=2048, reps=5):
    """Device copy microbenchmark beside the 8 TB/s vendor figure (report both): a 16-byte-per-lane
    device-to-device copy of `mib` MiB (torch's copy kernel on the context's stream, far larger than the 256 MiB
    Infinity Cache), HIP events on that stream; GB/s counts bytes read + bytes written.  Outside every timed region."""
    n = mib << 20
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.random_(0, 255)
    b.copy_(a)
    ctx.sync()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()          # (torch's current stream: the stream its copy kernel runs on)
        b.copy_(a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del a, b
    torch.cuda.empty_cache()
    return {"GBps": round(2 * n / (best * 1e-3) / 1e9, 1), "bytes_copied": n, "ms": round(best, 4),
            "how": "torch device copy (16 B per lane), read + written bytes / best of %d" % reps}


def zstd_object(ctx, torch):
    """SURVEY 8f-3 in the same driver-run command: 16 384 zstd frames of 64 KiB (libzstd level 3 output of synthetic
    text-like data, 64 distinct frames tiled) through la_gpu_zstd_decode, inputs resident in HBM, every frame's status
    and the first 64 frames' bytes checked; libzstd on one host core beside it.  Carried as `tertiary`."""
    import ctypes
    import random
    import zstd_support as Z
    from libarchive_amd import zstd as LZ
    z = Z.libzstd()
    if z is None:
        return {"skipped": "no libzstd.so.1 in this image to make the synthetic stream"}
    nfr, kib, lvl = 16384, 64, 3
    rnd = random.Random(1)
    uniq = []
    for i in range(64):
        d = Z.gen(rnd, kib * 1024, 2 if i % 2 else 4)
        uniq.append((d, Z.zstd_compress(z, d, lvl)))
    img = b"".join(uniq[i % 64][1] for i in range(nfr))
    plain_len = nfr * kib * 1024
    # CPU: the library the reference's filter calls, one core, the 64 distinct frames in one ZSTD_decompress call
    sample = b"".join(u[1] for u in uniq)
    buf = ctypes.create_string_buffer(64 * kib * 1024 + 64)
    t0 = time.time()
    reps = 0
    while time.time() - t0 < 5.0:
        z.ZSTD_decompress(buf, len(buf), sample, len(sample))
        reps += 1
    cpu_s = time.time() - t0
    cpu = reps * 64 * kib * 1024 / cpu_s / 2**20
    frames, end_kind, consumed, dst_bytes = LZ.index_image(img)
    assert len(frames) == nfr and consumed == len(img)
    d_src = torch.from_numpy(np.frombuffer(img, dtype=np.uint8).copy()).cuda()
    plan = LZ.ZstdDevicePlan(ctx, d_src, frames, dst_bytes)
    plan.run()
    res = plan.results()
    ok = bool((res["status"] == 0).all()) and int(res["out_len"].sum()) == plain_len
    tile = 64 * kib * 1024
    head = plan.d_dst[:tile].cpu().numpy().tobytes()
    ok = ok and head == b"".join(u[0] for u in uniq)
    # every tile of 64 frames must equal the first (the frames carry no content checksum: status and summed lengths
    # alone would not show a lane- or wave-scheduling bug that damages later tiles)
    ntile = plain_len // tile
    ok = ok and bool(torch.equal(plan.d_dst[:ntile * tile].view(ntile, tile), plan.d_dst[:tile].unsqueeze(0).expand(ntile, tile)))
    for _ in range(2):
        plan.run()
    ctx.sync()
    K = 5
    t0 = time.time()
    for _ in range(K):
        plan.run()
    ctx.sync()
    dt = (time.time() - t0) / K
    achieved = (len(img) + plain_len) / dt / 1e9
    return {
        "metric": "decompressed MiB/s (whole node), zstd read filter data plane, bit-exact",
        "value": round(plain_len / dt / 2**20, 1), "unit": "MiB/s", "n_gpus": 1, "steps": K, "warmup": 3,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d zstd frames of %d KiB decoded (libzstd %d level %d output, 64 distinct frames tiled), "
                               "one wave per frame, inputs resident in HBM" % (nfr, kib, z.ZSTD_versionNumber(), lvl),
                   "compressed_bytes": len(img), "decoded_bytes": plain_len},
        "bit_exact": ok,
        "roofline": {"bound": "hbm", "kernel": "zstd_frames_wave_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes": len(img) + plain_len},
        "cpu_baseline": {"value": round(cpu, 1), "unit": "MiB/s", "cores": 1, "kind": "libzstd",
                         "sample": "%d x %d MiB decoded, ZSTD_decompress over the 64 distinct frames, %.1f s" % (reps, 64 * kib // 1024, cpu_s)},
    }

def gz_traffic(args):
    """PMC-measured HBM bytes of the two inflate kernels per step for the C3 workload (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes, tools/exp_traffic_gz.sh, committed as profiles/r03_traffic_gzip.json); quoted only when the file was
    measured on the library in use and the workload is the full C3 size."""
    tpath = os.path.join(ROOT, "profiles", "r03_traffic_gzip.json")
    if not os.path.exists(tpath) or abs(args.gib - 16.0) > 1e-6:
        return None
    rec = json.load(open(tpath))
    if rec.get("library_sha16") != kernel_source_id():
        return None
    return rec.get("inflate_symbols_plus_expand", {}).get("hbm_bytes_per_step")


def main_gzip(args, as_secondary=False):
    """configs[2] shape: concatenated gzip members of 64 KiB with a BGZF-style size subfield,
    CRC32 + ISIZE verified on the device.  `--workload gzip` prints it as its own line; the default run
    (the lz4 headline) carries it as the `secondary` object of its line, so that C3 is timed in the same
    driver-run command."""
    # The host side first -- stream synthesis and member compression -- with THREADS (zlib releases
    # the GIL) and before anything here initialises the GPU: a forked worker of a process that holds an
    # HSA / profiler state crashes in the profiler's signal handler at pool teardown (round-1 abort under
    # rocprofv3 --pmc).
    from concurrent.futures import ThreadPoolExecutor
    import streams as S
    uniq_mib = min(args.unique_mib, 256)
    frames = (uniq_mib << 20) // (BPF * BLOCK)
    _, plain = S.synth_lz4_stream(SEED, 0, frames, BPF, BLOCK, nthreads=min(16, os.cpu_count() or 1))
    pieces = [(plain[i:i + BLOCK].tobytes(),) for i in range(0, plain.size, BLOCK)]
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
        members = list(pool.map(_gz_make_member, pieces, chunksize=64))
    import torch
    import libarchive_amd as la
    from libarchive_amd import _native as N
    from libarchive_amd.gzip import GzDevicePlan

    assert torch.cuda.is_available()
    device = torch.device("cuda", 0)
    ctx = la.GpuContext(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    img = np.frombuffer(b"".join(members), dtype=np.uint8)
    idx = la.gz_index(img, at_eof=True)
    assert len(idx.members) == len(pieces) and idx.speculative == 0
    tiles = max(1, int(round(args.gib * (1 << 30) / plain.size)))
    nm, clen = len(idx.members), int(img.size)
    d_src = torch.from_numpy(img.copy()).to(device).repeat(tiles)
    mem = np.tile(idx.members, tiles)
    t_of = np.repeat(np.arange(tiles, dtype=np.uint64), nm)
    mem["src_off"] += t_of * np.uint64(clen)
    mem["dst_off"] += t_of * np.uint64(idx.max_out)
    tiled = N.GzIndex(mem, None, idx.end_kind, clen * tiles, idx.max_out * tiles, 0)
    plan = GzDevicePlan(ctx, d_src, tiled)
    C_bytes, U_bytes = int(d_src.numel()), int(tiled.max_out)
    for _ in range(args.warmup):
        plan.run(args.extra_options)
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    phase_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(args.extra_options)
        for name, ms in ctx.profile_read():
            phase_ms.setdefault(name, []).append(ms)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sm = plan.summary()
    ok = int(sm["n_bad_units"]) == 0 and int(sm["total_out"]) == U_bytes
    ok = ok and bool(torch.equal(plan.d_dst[:plain.size].cpu(), torch.from_numpy(plain)))
    for t in range(1, tiles):
        ok = ok and bool(torch.equal(plan.d_dst[:plain.size], plan.d_dst[t * plain.size:(t + 1) * plain.size]))
    cpu = None
    if not args.no_cpu_baseline:
        import oracle_lib as O
        reps, el, t1 = 0, 0.0, time.time()
        while el < 10.0:
            out, res = O.gzip_stream_decode(img, plain.size + 64)
            reps += 1
            el = time.time() - t1
        assert res.rc == 0 and np.array_equal(out, plain)
        cpu = dict(value=round(reps * plain.size / el / (1 << 20), 1), unit="MiB/s", cores=1, kind="port",
                   sample="%d x %d MiB decoded, oracle gzip filter (inflate + trailer CRC32), %.1f s" % (reps, plain.size >> 20, el))
        # the box's own zlib behind the product walker's member table: what libarchive's filter executes (gzip.c:479)
        import threading
        import zlib
        raw = img.tobytes()
        spans = [(int(m["src_off"]), int(m["src_len"])) for m in idx.members]

        def zl_pass():
            n = 0
            for o, ln in spans:
                n += len(zlib.decompressobj(-15).decompress(raw[o:o + ln]))
            return n
        assert zl_pass() == plain.size
        reps, el, t1 = 0, 0.0, time.time()
        while el < 5.0:
            zl_pass()
            reps += 1
            el = time.time() - t1
        variants = [dict(value=round(reps * plain.size / el / (1 << 20), 1), unit="MiB/s", cores=1, kind="zlib",
                         sample="%d x %d MiB, zlib %s inflate per member (no CRC check, like the reference), %.1f s" % (reps, plain.size >> 20, zlib.ZLIB_RUNTIME_VERSION, el))]
        ncores = os.cpu_count() or 1
        counts, stop = [0] * ncores, time.time() + 5.0

        def replica(i):
            while time.time() < stop:
                zl_pass()
                counts[i] += 1
        t1 = time.time()
        th = [threading.Thread(target=replica, args=(i,)) for i in range(ncores)]
        [t.start() for t in th]
        [t.join() for t in th]
        el = time.time() - t1
        variants.append(dict(value=round(sum(counts) * plain.size / el / (1 << 20), 1), unit="MiB/s", cores=ncores, kind="zlib replicas",
                             sample="%d independent replicas on all host cores, %d passes of %d MiB in %.1f s" % (ncores, sum(counts), plain.size >> 20, el)))
        cpu["variants"] = variants
    # deflate decode = entropy decode (inflate_symbols) + LDS-window expand (inflate_expand) + in-place
    # kernel for members the window kernel cannot take (inflate); older single-kernel paths only have the last
    inf_ms = float(sum(np.mean(phase_ms[k]) for k in ("inflate_symbols", "inflate_expand", "inflate") if k in phase_ms))
    ach = (C_bytes + U_bytes) / (inf_ms * 1e-3) / 1e9
    line = {
        "metric": "decompressed MiB/s (whole node), gzip filter, CRC32 verified, bit-exact",
        "value": round(U_bytes * args.steps / dt / (1 << 20), 1), "unit": "MiB/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "C3 shape: raw-format + gzip filter, %.2f GiB decoded, %d BGZF-style 64 KiB members (zlib level 6), CRC32+ISIZE verified"
                               % (U_bytes / (1 << 30), plan.n),
                   "compressed_bytes_per_gpu": C_bytes, "decoded_bytes_per_gpu": U_bytes},
        "bit_exact": bool(ok), "phases_ms": {k: round(float(np.mean(v)), 3) for k, v in phase_ms.items()},
        "roofline": {"bound": "hbm", "kernel": "inflate (symbols + expand)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": gz_traffic(args),
                     "algorithmic_bytes": C_bytes + U_bytes},
        "cpu_baseline": cpu}
    ctx.close()
    del plan, d_src
    torch.cuda.empty_cache()
    if as_secondary:
        return line
    print(json.dumps(line), flush=True)
    if not ok:
        sys.exit(3)


def launch_ranks(args):
    """--gpus N without a launcher: start N ranks as a child (torch.distributed.run), from a process
    that has not touched a GPU, and leave with the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.workload == "gzip":
        return main_gzip(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch %d ranks (python bench.py --gpus %d starts them itself)"
                 % (args.gpus, world, args.gpus, args.gpus))
    import streams as S
    api = None
    if world == 1 and not args.no_cpu_baseline:
        api = api_level(S, args)      # child process, before this one initialises the GPU
    import torch
    import torch.distributed as dist
    import libarchive_amd as la
    from libarchive_amd import _native as N
    from libarchive_amd.lz4 import Lz4DevicePlan

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # LA_BENCH_BACKEND=gloo is the REHEARSAL of the N > 1 path on a box with fewer GPUs than ranks: the ranks
    # share the devices there are and the collectives run on CPU tensors.  The measured configuration is nccl
    # (= RCCL over xGMI), one GPU per rank; a rehearsal line says so in `config`.
    backend = os.environ.get("LA_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        sys.exit("bench.py: %d ranks but %d GPU(s) visible" % (world, ndev))
    local = local % ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    coll_device = device if backend == "nccl" else torch.device("cpu")

    ctx = la.GpuContext(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    img_u, idx_u, d_src, tiled, info = build_tiled_stream(torch, la, S, rank, world, args.unique_mib, args.gib, device)
    plan = Lz4DevicePlan(ctx, d_src, tiled)
    C_bytes = int(d_src.numel())
    U_bytes = int(tiled.max_out)  # every synthetic block decodes to exactly 64 KiB
    opts = (N.LA_LZ4_OPT_GENERAL_ONLY if args.general_only else 0) | args.extra_options

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.run(opts)
    barrier()

    ctx.profile_enable(True)
    phase_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(opts)
        for name, ms in ctx.profile_read():  # waits for this step's last event only
            phase_ms.setdefault(name, []).append(ms)
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)

    # ---- verification outside the timed region: checksums-of-everything + oracle sample ----
    sm = plan.summary()
    ok = (int(sm["total_out"]) == U_bytes and int(sm["n_bad_units"]) == 0 and int(sm["n_bad_frames"]) == 0)
    # the stream is the unique region repeated: the decoded range must repeat with the region's period
    tile_bytes = info["frames_unique"] * BPF * BLOCK
    for o in range(0, U_bytes - tile_bytes, tile_bytes):
        n = min(tile_bytes, U_bytes - tile_bytes - o)
        ok = ok and bool(torch.equal(plan.d_dst[o:o + n], plan.d_dst[o + tile_bytes:o + tile_bytes + n]))

    # the only exchange: per-rank summaries (no decoded bytes move; outputs stay sharded)
    from libarchive_amd.shard import exchange_summaries
    dt_max, U_all, C_all, ok_all = exchange_summaries(dist, coll_device, dt, U_bytes, C_bytes, ok)

    # explicit "single stream split" mode: the decoded ranges gathered to rank 0 over RCCL, timed on its own
    gather = None
    if world > 1 and args.gather:
        from libarchive_amd.shard import gather_ranges_into
        barrier()
        t0 = time.perf_counter()
        buf, slot, sizes = gather_ranges_into(dist, coll_device, plan.d_dst[:U_bytes].to(coll_device), root=0)
        barrier()
        dt_g = time.perf_counter() - t0
        if rank == 0:
            g_ok = all(bool(torch.equal(buf[r * slot:r * slot + min(sizes[r], tile_bytes)][-4096:],
                                        buf[:min(sizes[r], tile_bytes)][-4096:])) for r in range(world)
                       if info["first_tile_frame"] == 0 and sizes[r] >= tile_bytes)
            step_s = dt_max / args.steps
            gather = {"gather_ms": round(dt_g * 1e3, 2), "bytes_to_root": int(sum(sizes) - sizes[0]),
                      "GBps_into_root": round((sum(sizes) - sizes[0]) / dt_g / 1e9, 1),
                      "value_with_gather": round(U_all / (step_s + dt_g) / (1 << 20), 1), "unit": "MiB/s",
                      "spot_check": bool(g_ok)}
        del buf

    if rank == 0:
        cpu = None
        sample_ok = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is an N=1 measurement
            import oracle_lib as O  # checker / baseline only
            cpu, ref_out = cpu_baseline(S, O, img_u, idx_u, args.cpu_sample_mib)
            got = plan.d_dst[:len(ref_out)].cpu().numpy()
            sample_ok = bool(np.array_equal(got, ref_out))
            ok_all = ok_all and sample_ok
        exp_ms = float(np.mean(phase_ms.get("lz4_expand", [float("nan")])))
        # PMC-measured HBM bytes of the expand kernel for this exact workload (separate rocprofv3 --pmc
        # passes, tools/exp_traffic.sh, committed under profiles/), per slice launch like `achieved`.
        # The file names the kernel source it was measured on: a stale file is refused, not quoted.
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if os.path.exists(tpath) and abs(args.gib - 16.0) < 1e-6 and not args.general_only and not args.extra_options:
            rec = json.load(open(tpath))
            if rec.get("library_sha16") == kernel_source_id():
                traffic = rec["lz4_expand_fast_kernel"]["hbm_bytes_per_launch"]
            else:
                traffic_note = "profiles/r03_traffic.json was measured on another build of libla_gpu.so (%s, now %s): not quoted" % (
                    rec.get("library_sha16"), kernel_source_id())
        copy_peak = measured_copy_peak(ctx, torch) if world == 1 else None
        nl = 4 if (plan.n_blocks >= 32768 and not args.general_only) else 1   # slice launches of the expand kernel per step
        # per-launch algorithmic bytes / per-launch duration (the slices are equal, so this is the ratio of the sums)
        achieved = (C_bytes + U_bytes) / (exp_ms * 1e-3) / 1e9
        step_ms = dt_max / args.steps * 1e3
        line = {
            "metric": "decompressed MiB/s (whole node), lz4 filter, XXH32 verified, bit-exact",
            "value": round(U_all * args.steps / dt_max / (1 << 20), 1),
            "unit": "MiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "C2: raw-format + lz4 filter, %.2f GiB decoded per GPU, %d independent 64 KiB blocks per GPU in frames of 16, block+content+header XXH32 verified"
                            % (U_bytes / (1 << 30), plan.n_blocks),
                "compressed_bytes_per_gpu": C_bytes,
                "decoded_bytes_per_gpu": U_bytes,
                "unique_region_mib": args.unique_mib,
                "tiles": info["tiles"],
                "parallelism": "ONE stream of %d x %.0f GiB cut on frame boundaries into %d ranges balanced by C+U, one range per GPU, outputs stay sharded"
                               % (world, args.gib, world),
                "rank0_frame_range": list(info["frame_range"]),
                "collectives": "RCCL (nccl backend)" if backend == "nccl" else "REHEARSAL: %s on CPU tensors, %d rank(s) sharing %d GPU(s) -- not a scaling measurement" % (backend, world, ndev),
                "expand_kernel": "general" if args.general_only else "auto",
                "host_index_ms_per_gib_compressed": round(info["index_s_unique"] * 1e3 / (info["unique_compressed"] / (1 << 30)), 2),
            },
            "bit_exact": bool(ok_all),
            "oracle_sample_match": sample_ok,
            "phases_ms": {k: round(float(np.mean(v)), 3) for k, v in phase_ms.items()},
            "roofline": {
                "bound": "hbm",
                "kernel": "lz4_expand",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "launches_per_step": nl,
                "algorithmic_bytes": (C_bytes + U_bytes) // nl,   # per launch, like traffic and achieved
                "traffic_note": traffic_note,
                "launch_ms": round(exp_ms / nl, 4),
                "whole_step_frac": round((C_bytes + U_bytes) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # measured on this box beside the vendor peak (`peak`, which `frac` is quoted against)
                "peak_measured_copy": copy_peak,
                "frac_of_measured_copy": round(achieved / copy_peak["GBps"], 4) if copy_peak else None,
            },
            "cpu_baseline": cpu,
            "api_level": api,
            "gather_mode": gather,
        }
        if world == 1 and not args.no_secondary and not args.no_cpu_baseline and abs(args.gib - 16.0) < 1e-6:
            # BASELINE configs[2] (C3) in the same command: the lz4 buffers go first
            del plan, d_src
            torch.cuda.empty_cache()
            line["secondary"] = main_gzip(args, as_secondary=True)
            ok_all = ok_all and bool(line["secondary"]["bit_exact"])
            torch.cuda.empty_cache()
            line["tertiary"] = zstd_object(ctx, torch)
            ok_all = ok_all and bool(line["tertiary"].get("bit_exact", True))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if not ok_all and rank == 0:
        sys.exit(3)


if __name__ == "__main__":
    main()
