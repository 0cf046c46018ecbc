"""Device-resident gzip batch decode through the C ABI (harness for tests / bench.py).
Plumbing only: torch allocates the HBM buffers; all work is la_gpu_gzip_decode()."""
import numpy as np

from . import _native as N


class GzDevicePlan:
    def __init__(self, ctx, d_src, index, device=None):
        import torch
        dev = d_src.device if device is None else device
        self.ctx, self.index, self.d_src = ctx, index, d_src
        n = len(index.members)
        self.n = n
        self.dst_cap = int(index.max_out)
        self.d_members = torch.from_numpy(index.members.view(np.uint8).reshape(-1).copy()).to(dev)
        self.d_dst = torch.empty(max(self.dst_cap, 16), dtype=torch.uint8, device=dev)
        self.d_results = torch.zeros(max(n, 1) * N.GZ_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.d_summary = torch.zeros(N.SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        b = N._GzBatchC()
        b.d_src = d_src.data_ptr(); b.src_bytes = d_src.numel()
        b.d_members = self.d_members.data_ptr(); b.n_members = n
        b.d_dst = self.d_dst.data_ptr(); b.dst_cap = self.dst_cap
        b.d_results = self.d_results.data_ptr(); b.d_summary = self.d_summary.data_ptr()
        self.batch = b

    def run(self, options=0):
        self.batch.options = options
        self.ctx.gzip_decode(self.batch)

    def summary(self):
        self.ctx.sync()
        return self.d_summary.cpu().numpy().view(N.SUMMARY_DTYPE)[0]

    def results(self):
        self.ctx.sync()
        return self.d_results.cpu().numpy().view(N.GZ_RESULT_DTYPE)[:self.n]


def compress_to_members(ctx, d_plain, chunk_bytes=49152, mtime=0):
    """Device gzip compression (la_gpu_gzip_compress): d_plain is a 1-D uint8 CUDA tensor; returns a uint8 CUDA
    tensor holding the concatenated gzip members (harness for the tests)."""
    import torch
    from . import _native as N
    n = int(d_plain.numel())
    cap = int(N.gpu_lib().la_gpu_gzip_compress_bound(n, chunk_bytes))
    d_out = torch.empty(max(cap, 16), dtype=torch.uint8, device=d_plain.device)
    d_len = torch.zeros(1, dtype=torch.int64, device=d_plain.device)
    b = N._GzcBatchC()
    b.d_src = d_plain.data_ptr() if n else None
    b.src_bytes = n
    b.chunk_bytes, b.mtime = chunk_bytes, mtime
    b.d_out, b.out_cap, b.d_out_bytes = d_out.data_ptr(), cap, d_len.data_ptr()
    ctx.gzip_compress(b)
    ctx.sync()
    total = int(d_len.cpu()[0])
    assert total <= cap, (total, cap)
    return d_out[:total]
