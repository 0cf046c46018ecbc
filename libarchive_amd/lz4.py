"""Device-resident LZ4 batch decode through the C ABI (harness for tests / bench.py).

Plumbing only: torch allocates the HBM buffers and provides the stream; all
work is done by la_gpu_lz4_decode() (include/la_gpu.h).  The stream-order event
resolution below mirrors what the C filter does in host/la_filter_lz4.c.
"""
import ctypes as C

import numpy as np

from . import _native as N


def _torch():
    import torch
    return torch


class Lz4DevicePlan:
    """HBM-resident job tables and outputs for one batch (built outside any timed region)."""

    def __init__(self, ctx, d_src, index, dst_cap=None, device=None):
        torch = _torch()
        dev = d_src.device if device is None else device
        self.ctx = ctx
        self.index = index
        self.d_src = d_src
        nb, nf = len(index.blocks), len(index.frames)
        self.n_blocks, self.n_frames = nb, nf
        if dst_cap is None:
            dst_cap = int(index.max_out)
        self.dst_cap = int(dst_cap)
        u8 = torch.uint8
        self.d_blocks = torch.from_numpy(index.blocks.view(np.uint8).reshape(-1).copy()).to(dev)
        self.d_frames = torch.from_numpy(index.frames.view(np.uint8).reshape(-1).copy()).to(dev)
        self.d_dst = torch.empty(max(self.dst_cap, 16), dtype=u8, device=dev)
        self.d_out_len = torch.zeros(max(nb, 1), dtype=torch.int32, device=dev)
        self.d_dst_off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
        self.d_block_status = torch.zeros(max(nb, 1), dtype=torch.int32, device=dev)
        self.d_frame_status = torch.zeros(max(nf, 1), dtype=torch.int32, device=dev)
        self.d_summary = torch.zeros(N.SUMMARY_DTYPE.itemsize, dtype=u8, device=dev)
        ctx.reserve(N.gpu_lib().la_gpu_lz4_workspace_bytes(nb, d_src.numel()))
        b = N._Lz4BatchC()
        b.d_src = d_src.data_ptr()
        b.src_bytes = d_src.numel()
        b.d_blocks = self.d_blocks.data_ptr() if nb else None
        b.n_blocks = nb
        b.d_frames = self.d_frames.data_ptr() if nf else None
        b.n_frames = nf
        b.d_dst = self.d_dst.data_ptr()
        b.dst_cap = self.dst_cap
        b.d_out_len = self.d_out_len.data_ptr()
        b.d_dst_off = self.d_dst_off.data_ptr()
        b.d_block_status = self.d_block_status.data_ptr()
        b.d_frame_status = self.d_frame_status.data_ptr()
        b.d_summary = self.d_summary.data_ptr()
        self.batch = b

    def run(self, options=0):
        """Enqueue one decode of the whole batch on the context's stream (no sync)."""
        self.batch.options = options
        self.ctx.lz4_decode(self.batch)

    def summary(self):
        self.ctx.sync()
        return self.d_summary.cpu().numpy().view(N.SUMMARY_DTYPE)[0]

    def arrays(self):
        """(out_len, dst_off, block_status, frame_status) as numpy, after a sync."""
        self.ctx.sync()
        return (self.d_out_len.cpu().numpy().view(np.uint32)[:self.n_blocks],
                self.d_dst_off.cpu().numpy().view(np.uint64),
                self.d_block_status.cpu().numpy().view(np.uint32)[:self.n_blocks],
                self.d_frame_status.cpu().numpy().view(np.uint32)[:self.n_frames])

    def resolve(self):
        """Stream-order outcome: (delivered_bytes, rc, message) exactly as the
        reference filter would report for the same image (rc 0 or -30)."""
        out_len, dst_off, bst, fst = self.arrays()
        return resolve_events(self.index, out_len, dst_off, bst, fst)

    def output(self, nbytes=None):
        self.ctx.sync()
        n = int(self.summary()["total_out"]) if nbytes is None else int(nbytes)
        return self.d_dst[:n]


def resolve_events(index, out_len, dst_off, bst, fst):
    """First event in stream order decides the outcome (failure detection).

    Per frame: header check byte (lz4.c:446-451) -> each block (checksum lz4.c:517-526,
    decode :594-598, a block of 0 bytes ends the stream, F11 i) -> content checksum
    (:639-662); after the last indexed unit: the walker's end kind."""
    ARCHIVE_FATAL = -30
    frames = index.frames
    for fi in range(len(frames)):
        f = frames[fi]
        first, n = int(f["first_block"]), int(f["n_blocks"])
        if fst[fi] == 3:  # LA_ST_LZ4_BAD_HEADER_SUM
            return int(dst_off[first]), ARCHIVE_FATAL, N.status_message(3)
        sl = slice(first, first + n)
        bad = np.nonzero((bst[sl] != 0) | (out_len[sl] == 0))[0]
        if bad.size:
            b = first + int(bad[0])
            if bst[b] != 0:
                return int(dst_off[b]), ARCHIVE_FATAL, N.status_message(int(bst[b]))
            return int(dst_off[b]), 0, ""
        if fst[fi] == 4:  # LA_ST_LZ4_BAD_CONTENT_SUM
            return int(dst_off[first + n]), ARCHIVE_FATAL, N.status_message(4)
    total = int(dst_off[len(index.blocks)])
    if index.end_kind in (N.LA_END_TRUNCATED, N.LA_END_MALFORMED, N.LA_END_MALFORMED_SKIP):
        return total, ARCHIVE_FATAL, N.end_message(index.end_kind)
    return total, 0, ""


def decode_image(ctx, image, device="cuda:0", options=0):
    """Convenience for tests: host image -> (decoded numpy bytes, rc, message).

    Uploads the image, runs the device batch, resolves the outcome in stream order."""
    torch = _torch()
    if isinstance(image, (bytes, bytearray, memoryview)):
        image = np.frombuffer(bytes(image), dtype=np.uint8)
    image = np.ascontiguousarray(image, dtype=np.uint8)
    idx = N.lz4_index(image, at_eof=True)
    if image.size:
        d_src = torch.from_numpy(image.copy()).to(device)
    else:
        d_src = torch.zeros(16, dtype=torch.uint8, device=device)[:0]
    plan = Lz4DevicePlan(ctx, d_src, idx)
    plan.run(options)
    delivered, rc, msg = plan.resolve()
    out = plan.d_dst[:delivered].cpu().numpy()
    return out, rc, msg, plan


def compress_to_frames(ctx, d_plain, block_size=65536, blocks_per_frame=16, flags=None):
    """Device LZ4 compression (la_gpu_lz4_compress): d_plain is a 1-D uint8 CUDA tensor; returns a uint8
    CUDA tensor holding the concatenated frames (harness for the tests and for stream synthesis)."""
    torch = _torch()
    if flags is None:
        flags = N.LA_LZ4C_BLOCK_SUM | N.LA_LZ4C_CONTENT_SUM
    n = int(d_plain.numel())
    cap = int(N.gpu_lib().la_gpu_lz4_compress_bound(n, block_size, blocks_per_frame))
    d_out = torch.empty(max(cap, 16), dtype=torch.uint8, device=d_plain.device)
    d_len = torch.zeros(1, dtype=torch.int64, device=d_plain.device)
    b = N._Lz4cBatchC()
    b.d_src = d_plain.data_ptr() if n else None
    b.src_bytes = n
    b.block_size, b.blocks_per_frame, b.flags = block_size, blocks_per_frame, flags
    b.d_out, b.out_cap, b.d_out_bytes = d_out.data_ptr(), cap, d_len.data_ptr()
    ctx.lz4_compress(b)
    ctx.sync()
    total = int(d_len.cpu()[0])
    assert total <= cap, (total, cap)
    return d_out[:total]
