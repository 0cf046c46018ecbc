"""Device-resident zstd batch decode through the C ABI (harness for tests / tools).  Plumbing only: torch allocates
the HBM buffers; the frame table comes from the host walker (la_zstd_index_build), all work is la_gpu_zstd_decode()."""
import ctypes as C

import numpy as np

from . import _native as N

ZSTD_FRAME_DTYPE = np.dtype([("src_off", "<u8"), ("src_len", "<u8"), ("dst_off", "<u8"), ("dst_cap", "<u8")])
ZSTD_RESULT_DTYPE = np.dtype([("status", "<u4"), ("reserved", "<u4"), ("out_len", "<u8")])


class _IndexResultC(C.Structure):
    _fields_ = [("n_frames", C.c_uint32), ("end_kind", C.c_int), ("consumed", C.c_uint64), ("dst_bytes", C.c_uint64),
                ("window_full", C.c_int)]


class _ZstdBatchC(C.Structure):
    _fields_ = [("d_src", C.c_void_p), ("src_bytes", C.c_uint64), ("d_frames", C.c_void_p), ("n_frames", C.c_uint32),
                ("options", C.c_uint32), ("d_dst", C.c_void_p), ("dst_cap", C.c_uint64), ("d_results", C.c_void_p)]


def index_image(image, at_eof=True, cap=1 << 20, out_budget=0, full=False):
    """(frames ndarray, end_kind, consumed, dst_bytes) of a host image (bytes / uint8 ndarray)."""
    lib = N.host_lib()
    buf = np.frombuffer(image, dtype=np.uint8) if not isinstance(image, np.ndarray) else image
    frames = np.zeros(cap, dtype=ZSTD_FRAME_DTYPE)
    res = _IndexResultC()
    lib.la_zstd_index_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(_IndexResultC)]
    lib.la_zstd_index_build(buf.ctypes.data, buf.size, 1 if at_eof else 0, int(out_budget), frames.ctypes.data, cap, C.byref(res))
    if full:
        return frames[:res.n_frames].copy(), res
    return frames[:res.n_frames].copy(), res.end_kind, res.consumed, res.dst_bytes


class ZstdDevicePlan:
    def __init__(self, ctx, d_src, frames, dst_bytes):
        import torch
        dev = d_src.device
        self.ctx, self.n = ctx, len(frames)
        self.d_frames = torch.from_numpy(frames.view(np.uint8).reshape(-1).copy()).to(dev)
        self.d_dst = torch.empty(max(int(dst_bytes), 16), dtype=torch.uint8, device=dev)
        self.d_results = torch.zeros(max(self.n, 1) * ZSTD_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        b = _ZstdBatchC()
        b.d_src, b.src_bytes = d_src.data_ptr(), d_src.numel()
        b.d_frames, b.n_frames = self.d_frames.data_ptr(), self.n
        b.d_dst, b.dst_cap = self.d_dst.data_ptr(), int(dst_bytes)
        b.d_results = self.d_results.data_ptr()
        self.batch = b
        lib = N.gpu_lib()
        lib.la_gpu_zstd_decode.argtypes = [C.c_void_p, C.POINTER(_ZstdBatchC)]
        lib.la_gpu_zstd_workspace_bytes.restype = C.c_uint64
        self._lib = lib

    def run(self, options=0):
        self.batch.options = options
        rc = self._lib.la_gpu_zstd_decode(self.ctx._h, C.byref(self.batch))
        if rc != 0:
            raise RuntimeError("la_gpu_zstd_decode: %d" % rc)

    def results(self):
        self.ctx.sync()
        return self.d_results.cpu().numpy().view(ZSTD_RESULT_DTYPE)[:self.n]
