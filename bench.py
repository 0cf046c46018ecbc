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
Independent blocks shard across ranks (each rank owns its own frame range: weak
scaling, no data-path collective); ranks exchange only their 32-byte batch summary.

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
    ap.add_argument("--general-only", action="store_true", help="force the general expand kernel")
    ap.add_argument("--extra-options", type=int, default=0, help="diagnostic: extra LA_LZ4_OPT_* bits")
    ap.add_argument("--workload", choices=["lz4", "gzip"], default="lz4",
                    help="lz4 = BASELINE configs[1] (the headline); gzip = configs[2] shape (64 KiB BGZF-style members)")
    return ap.parse_args()


def build_tiled_stream(torch, la, S, rank, unique_mib, total_gib, device):
    """Host: generate + index the unique region.  Device: tile stream and tables."""
    from libarchive_amd import _native as N
    frames_unique = max(1, (unique_mib << 20) // (BPF * BLOCK))
    tiles = max(1, int(round(total_gib * (1 << 30) / (frames_unique * BPF * BLOCK))))
    first_frame = rank * frames_unique  # every rank decodes its own frame range of the stream
    t0 = time.time()
    img, plain = S.synth_lz4_stream(SEED, first_frame, frames_unique, BPF, BLOCK,
                                    nthreads=min(16, os.cpu_count() or 1), want_plain=False)
    t_gen = time.time() - t0
    t0 = time.time()
    idx = la.lz4_index(img, at_eof=True)
    t_index = time.time() - t0
    assert idx.end_kind == N.LA_END_EOF and len(idx.frames) == frames_unique
    nb, nf, clen = len(idx.blocks), len(idx.frames), int(img.size)
    # device-side tiling of the stream and of the job tables
    d_unique = torch.from_numpy(img).to(device)
    d_src = d_unique.repeat(tiles)
    del d_unique
    blocks = np.tile(idx.blocks, tiles)
    frames = np.tile(idx.frames, tiles)
    t_of_b = np.repeat(np.arange(tiles, dtype=np.uint64), nb)
    t_of_f = np.repeat(np.arange(tiles, dtype=np.uint64), nf)
    blocks["src_off"] += t_of_b * np.uint64(clen)
    frames["desc_off"] += t_of_f * np.uint64(clen)
    frames["first_block"] += (t_of_f * np.uint64(nb)).astype(np.uint32)
    tiled = N.Lz4Index(blocks, frames, idx.end_kind, clen * tiles, idx.max_out * tiles)
    info = dict(frames_unique=frames_unique, tiles=tiles, gen_s=t_gen, index_s_unique=t_index,
                unique_compressed=clen, unique_blocks=nb)
    return img, idx, d_src, tiled, info


def cpu_baseline(S, O, img_unique, idx_unique, sample_mib, budget_s=12.0):
    """The oracle (port of the reference lz4 filter incl. all XXH32 checks), 1 thread."""
    nframes = len(idx_unique.frames)
    want = max(1, min(nframes, (sample_mib << 20) // (BPF * BLOCK)))
    end = int(idx_unique.frames["desc_off"][want] - 4) if want < nframes else int(img_unique.size)
    sample = img_unique[:end]
    cap = want * BPF * BLOCK + 64
    out, res = O.lz4_stream_decode(sample, cap)
    assert res.rc == 0 and len(out) == want * BPF * BLOCK
    # bounded sample: repeat the same frames until about `budget_s` of CPU work is done
    reps, dt, t0 = 0, 0.0, time.time()
    while dt < budget_s:
        out2, res = O.lz4_stream_decode(sample, cap)
        reps += 1
        dt = time.time() - t0
    return dict(value=round(reps * len(out) / dt / (1 << 20), 1), unit="MiB/s", cores=1, kind="port",
                sample="%d x %d MiB decoded (%d frames of the same stream), oracle lz4 filter with block+content XXH32 checks, %.1f s of CPU work"
                       % (reps, len(out) >> 20, want, dt)), out


def _gz_make_member(args):
    import struct, zlib
    data, = args
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9)
    body = co.compress(data) + co.flush()
    total = 18 + len(body) + 8
    hdr = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\x03" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1)
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def main_gzip(args):
    """configs[2] shape: concatenated gzip members of 64 KiB with a BGZF-style size subfield,
    CRC32 + ISIZE verified on the device.  Secondary line (the headline is the lz4 workload)."""
    import multiprocessing as mp
    import torch
    import libarchive_amd as la
    from libarchive_amd import _native as N
    from libarchive_amd.gzip import GzDevicePlan
    import streams as S

    assert torch.cuda.is_available()
    device = torch.device("cuda", 0)
    ctx = la.GpuContext(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    uniq_mib = min(args.unique_mib, 256)
    frames = (uniq_mib << 20) // (BPF * BLOCK)
    _, plain = S.synth_lz4_stream(SEED, 0, frames, BPF, BLOCK, nthreads=min(16, os.cpu_count() or 1))
    pieces = [(plain[i:i + BLOCK].tobytes(),) for i in range(0, plain.size, BLOCK)]
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:
        members = pool.map(_gz_make_member, pieces, chunksize=64)
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
    # deflate decode = entropy decode (inflate_symbols) + LDS-window expand (inflate_expand) + in-place
    # kernel for members the window kernel cannot take (inflate); older single-kernel paths only have the last
    inf_ms = float(sum(np.mean(phase_ms[k]) for k in ("inflate_symbols", "inflate_expand", "inflate") if k in phase_ms))
    ach = (C_bytes + U_bytes) / (inf_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "decompressed MiB/s (whole node), gzip filter, CRC32 verified, bit-exact",
        "value": round(U_bytes * args.steps / dt / (1 << 20), 1), "unit": "MiB/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "C3 shape: raw-format + gzip filter, %.2f GiB decoded, %d BGZF-style 64 KiB members (zlib level 6), CRC32+ISIZE verified"
                               % (U_bytes / (1 << 30), plan.n),
                   "compressed_bytes_per_gpu": C_bytes, "decoded_bytes_per_gpu": U_bytes},
        "bit_exact": bool(ok), "phases_ms": {k: round(float(np.mean(v)), 3) for k, v in phase_ms.items()},
        "roofline": {"bound": "hbm", "kernel": "inflate (symbols + expand)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes": C_bytes + U_bytes},
        "cpu_baseline": cpu}), flush=True)
    ctx.close()
    if not ok:
        sys.exit(3)


def main():
    args = parse()
    if args.workload == "gzip":
        return main_gzip(args)
    import torch
    import torch.distributed as dist
    import libarchive_amd as la
    from libarchive_amd import _native as N
    from libarchive_amd.lz4 import Lz4DevicePlan
    import streams as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    ctx = la.GpuContext(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    img_u, idx_u, d_src, tiled, info = build_tiled_stream(torch, la, S, rank, args.unique_mib, args.gib, device)
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
    tile_bytes = info["frames_unique"] * BPF * BLOCK
    out0 = plan.d_dst[:tile_bytes]
    for t in range(1, info["tiles"]):
        ok = ok and bool(torch.equal(out0, plan.d_dst[t * tile_bytes:(t + 1) * tile_bytes]))

    # the only exchange: per-rank summaries (no decoded bytes move; outputs stay sharded)
    from libarchive_amd.shard import exchange_summaries
    dt_max, U_all, C_all, ok_all = exchange_summaries(dist, device, dt, U_bytes, C_bytes, ok)

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
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and abs(args.gib - 16.0) < 1e-6 and not args.general_only:
            # PMC-measured HBM bytes of the expand kernel for this exact workload (separate
            # rocprofv3 --pmc passes, committed under profiles/), per slice launch like `achieved`
            traffic = json.load(open(tpath))["lz4_expand_fast_kernel"]["hbm_bytes_per_launch"]
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
                "parallelism": "blocks sharded over %d GPU(s), outputs stay sharded" % world,
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
                "launch_ms": round(exp_ms / nl, 4),
                "whole_step_frac": round((C_bytes + U_bytes) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if not ok_all and rank == 0:
        sys.exit(3)


if __name__ == "__main__":
    main()
