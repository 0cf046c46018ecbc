"""N > 1 path on CPU: two gloo ranks shard a synthetic many-frame stream by frame range,
each decodes ITS range (the oracle stands in for the device here -- this test is about the
sharding and the summary exchange, not the kernels), summaries are exchanged with the same
helper bench.py uses, and an explicit gather of decoded ranges reproduces the whole payload."""
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
from libarchive_amd.shard import exchange_summaries, gather_ranges, shard_range

FRAMES, BPF, BS = 7, 3, 4096


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 262144, 1000003):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, w, r) for r in range(w)]
            assert got[0][0] == 0 and sum(c for _, c in got) == n
            for (a, c), (b, _) in zip(got, got[1:]):
                assert a + c == b
            assert max(c for _, c in got) - min(c for _, c in got) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_range(FRAMES, world, rank)
    img, plain = S.synth_lz4_stream(11, first, count, BPF, BS, nthreads=1)
    idx = la.lz4_index(img)                      # product host walker on this rank's shard
    out, res = O.lz4_stream_decode(img, plain.size + 16)
    ok = res.rc == 0 and np.array_equal(out, plain) and len(idx.frames) == count
    dt, U, C, all_ok = exchange_summaries(dist, torch.device("cpu"), 0.5 + rank, plain.size, img.size, ok)
    whole = gather_ranges(dist, torch.device("cpu"), torch.from_numpy(out.copy()), root=0)
    if rank == 0:
        q.put((dt, U, C, all_ok, whole.numpy().tobytes()))
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
