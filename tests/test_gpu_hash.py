"""GPU parity for the hash kernels through the C ABI (la_gpu_xxh32_many / la_gpu_crc32_many)."""
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _run(ctx, fn, data, jobs):
    import torch
    from libarchive_amd import _native as N
    d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    j = np.zeros(len(jobs), dtype=N.HASH_JOB_DTYPE)
    for i, (off, ln, seed) in enumerate(jobs):
        j[i] = (off, ln, seed)
    dj = torch.from_numpy(j.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.zeros(len(jobs), dtype=torch.int32, device="cuda")
    fn(d.data_ptr(), dj.data_ptr(), len(jobs), out.data_ptr())
    ctx.sync()
    return out.cpu().numpy().view(np.uint32)


def _jobs(rnd, total):
    jobs = [(0, 0, 0), (3, 1, 5), (1, 15, 0), (2, 16, 9), (5, 17, 0), (0, total, 0)]
    for _ in range(300):
        ln = rnd.choice([rnd.randint(0, 64), rnd.randint(0, 5000), rnd.randint(0, total)])
        off = rnd.randint(0, total - ln)
        jobs.append((off, ln, rnd.getrandbits(32)))
    return jobs


def test_xxh32_many(gpu_ctx):
    rnd = random.Random(1)
    data = rnd.randbytes(300000)
    jobs = _jobs(rnd, len(data))
    got = _run(gpu_ctx, gpu_ctx.xxh32_many, data, jobs)
    for (off, ln, seed), g in zip(jobs, got):
        assert int(g) == O.xxh32(data[off:off + ln], seed), (off, ln, seed)


def test_crc32_many(gpu_ctx):
    rnd = random.Random(2)
    data = rnd.randbytes(300000)
    jobs = _jobs(rnd, len(data))
    got = _run(gpu_ctx, gpu_ctx.crc32_many, data, jobs)
    for (off, ln, seed), g in zip(jobs, got):
        want = zlib.crc32(data[off:off + ln], seed)
        assert int(g) == want == O.crc32(data[off:off + ln], seed), (off, ln, seed)
