"""N > 1 path on CPU: ONE synthetic many-frame stream is indexed once with the product's host
walker, cut into per-rank frame ranges balanced by C + U with the product's splitter (the one
bench.py uses), and every gloo rank takes only ITS byte range of the image.  No GPU exists
here, so the bytes of a range are decoded by the oracle (checker role only -- the same split
runs through the device kernels in tests/test_gpu_lz4.py::test_one_stream_split_over_ranks);
summaries are exchanged with the helper bench.py uses, and an explicit gather of decoded
ranges must reproduce the whole payload: the union of the ranges is the stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import streams as S
import libarchive_amd as la
from libarchive_amd.shard import (exchange_summaries, frame_weights, gather_ranges, gather_ranges_into,
                                  shard_range, slice_index, split_stream)

FRAMES, BPF, BS = 7, 3, 4096


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 262144, 1000003):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, w, r) for r in range(w)]
            assert got[0][0] == 0 and sum(c for _, c in got) == n
            for (a, c), (b, _) in zip(got, got[1:]):
                assert a + c == b
            assert max(c for _, c in got) - min(c for _, c in got) <= 1


def test_split_stream_cuts_on_frame_boundaries_and_balances():
    img, _ = S.synth_lz4_stream(5, 0, 40, 2, 4096, nthreads=1)
    idx = la.lz4_index(img)
    w, bounds = frame_weights(idx, img.size)
    assert int(bounds[0]) == 0 and int(bounds[-1]) == img.size and len(w) == 40
    for world in (1, 2, 3, 8, 40, 41):
        rs = split_stream(idx, img.size, world)
        assert rs[0][0] == 0 and rs[-1][1] == 40 and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        loads = [float(w[lo:hi].sum()) for lo, hi in rs]
        if world <= 8:      # no rank carries more than its share plus one frame
            assert max(loads) <= float(w.sum()) / world + float(w.max())
        covered = 0
        for lo, hi in rs:
            sub, a, b = slice_index(idx, img.size, lo, hi)
            fresh = la.lz4_index(img[a:b])          # walking the slice alone gives the same tables
            assert np.array_equal(fresh.blocks, sub.blocks) and np.array_equal(fresh.frames, sub.frames)
            covered += b - a
        assert covered == img.size


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img, plain = S.synth_lz4_stream(11, 0, FRAMES, BPF, BS, nthreads=1)      # the ONE stream
    idx = la.lz4_index(img)                                                  # indexed once (product walker)
    f_lo, f_hi = split_stream(idx, img.size, world)[rank]
    sub, b_lo, b_hi = slice_index(idx, img.size, f_lo, f_hi)
    mine = img[b_lo:b_hi]                                                    # the only bytes this rank touches
    want = plain[f_lo * BPF * BS:f_hi * BPF * BS]
    out, res = O.lz4_stream_decode(mine, want.size + 16)
    ok = res.rc == 0 and np.array_equal(out, want) and len(sub.frames) == f_hi - f_lo
    dt, U, C, all_ok = exchange_summaries(dist, torch.device("cpu"), 0.5 + rank, want.size, mine.size, ok)
    whole = gather_ranges(dist, torch.device("cpu"), torch.from_numpy(out.copy()), root=0)
    buf, slot, sizes = gather_ranges_into(dist, torch.device("cpu"), torch.from_numpy(out.copy()), root=0)
    if rank == 0:
        again = b"".join(buf[r * slot:r * slot + n].numpy().tobytes() for r, n in enumerate(sizes))
        q.put((dt, U, C, all_ok and again == whole.numpy().tobytes(), whole.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_exchange():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    dt, U, C, ok, whole = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full_img, full_plain = S.synth_lz4_stream(11, 0, FRAMES, BPF, BS, nthreads=1)
    assert ok and dt == 1.5 and U == full_plain.size and C == full_img.size
    assert whole == full_plain.tobytes()


def test_bench_refuses_a_gpus_world_size_mismatch():
    """`bench.py --gpus 2` inside a launcher that started ONE rank (WORLD_SIZE=1) must stop with the mismatch message
    before touching a GPU -- a line with n_gpus 2 timed on one rank would be a wrong scaling point."""
    import subprocess, sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in (p.stderr + p.stdout)
