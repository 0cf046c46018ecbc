"""ctypes bindings of the native libraries (libla_gpu.so, libla_host.so).

Mirrors include/la_gpu.h and include/la_host.h one to one.  No compute happens
in Python; a missing library raises NativeLibraryMissing (never a fallback).
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# LA_GPU_LIB selects a diagnostic build of the data plane (tools/exp_*.sh) without touching the shipped file
GPU_LIB_PATH = os.environ.get("LA_GPU_LIB") or os.path.join(_PKG, "csrc", "libla_gpu.so")
HOST_LIB_PATH = os.path.join(_PKG, "host", "libla_host.so")

LA_OK = 0

# struct la_lz4_block / la_lz4_frame / la_hash_job / la_batch_summary (include/la_gpu.h)
LZ4_BLOCK_DTYPE = np.dtype([("src_off", "<u8"), ("src_len", "<u4"), ("dst_cap", "<u4"),
                            ("flags", "<u4"), ("block_sum", "<u4")])
LZ4_FRAME_DTYPE = np.dtype([("desc_off", "<u8"), ("desc_len", "<u4"), ("first_block", "<u4"),
                            ("n_blocks", "<u4"), ("flags", "<u4"), ("content_sum", "<u4"),
                            ("reserved", "<u4")])
HASH_JOB_DTYPE = np.dtype([("off", "<u8"), ("len", "<u4"), ("seed", "<u4")])
SUMMARY_DTYPE = np.dtype([("total_out", "<u8"), ("n_bad_units", "<u4"), ("n_bad_frames", "<u4"),
                          ("first_bad_unit", "<u4"), ("first_bad_frame", "<u4"),
                          ("first_zero_unit", "<u4"), ("reserved", "<u4")])
GZ_MEMBER_DTYPE = np.dtype([("src_off", "<u8"), ("src_len", "<u4"), ("dst_cap", "<u4"), ("dst_off", "<u8")])
GZ_RESULT_DTYPE = np.dtype([("status", "<u4"), ("out_len", "<u4"), ("consumed", "<u4"), ("crc32", "<u4")])
assert LZ4_BLOCK_DTYPE.itemsize == 24 and LZ4_FRAME_DTYPE.itemsize == 32
assert HASH_JOB_DTYPE.itemsize == 16 and SUMMARY_DTYPE.itemsize == 32

LA_LZ4B_STORED, LA_LZ4B_CHECKSUM, LA_LZ4B_DEPENDENT, LA_LZ4B_FIRST = 1, 2, 4, 8
LA_LZ4F_CONTENT_SUM, LA_LZ4F_HEADER_SUM, LA_LZ4F_CONT, LA_LZ4F_OPEN, LA_LZ4F_HASHED = 1, 2, 4, 8, 16
LA_LZ4_OPT_GENERAL_ONLY, LA_LZ4_OPT_NO_VERIFY, LA_LZ4_OPT_PARSE_V1, LA_LZ4_OPT_EXPAND_INORDER = 1, 2, 4, 8
LA_GPU_ABI_VERSION = 3

(LA_END_EOF, LA_END_TRUNCATED, LA_END_MALFORMED, LA_END_MALFORMED_SKIP, LA_END_EMPTY_FRAME,
 LA_END_NEED_MORE, LA_END_GZ_NO_TRAILER, LA_END_GZ_TOO_LARGE) = range(8)


class NativeLibraryMissing(RuntimeError):
    pass


class _Lz4BatchC(C.Structure):
    _fields_ = [
        ("d_src", C.c_void_p), ("src_bytes", C.c_uint64),
        ("d_blocks", C.c_void_p), ("n_blocks", C.c_uint32),
        ("d_frames", C.c_void_p), ("n_frames", C.c_uint32),
        ("d_dst", C.c_void_p), ("dst_cap", C.c_uint64),
        ("d_out_len", C.c_void_p), ("d_dst_off", C.c_void_p),
        ("d_block_status", C.c_void_p), ("d_frame_status", C.c_void_p),
        ("d_summary", C.c_void_p),
        ("options", C.c_uint32), ("hist_len", C.c_uint32),
        ("d_carry_in", C.c_void_p), ("d_carry_out", C.c_void_p),   # content hash across batches (NULL here)
    ]


class _Lz4cBatchC(C.Structure):
    _fields_ = [
        ("d_src", C.c_void_p), ("src_bytes", C.c_uint64),
        ("block_size", C.c_uint32), ("blocks_per_frame", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32),
        ("d_out", C.c_void_p), ("out_cap", C.c_uint64), ("d_out_bytes", C.c_void_p),
    ]


LA_LZ4C_BLOCK_SUM, LA_LZ4C_CONTENT_SUM = 1, 2


class _GzcBatchC(C.Structure):
    _fields_ = [
        ("d_src", C.c_void_p), ("src_bytes", C.c_uint64),
        ("chunk_bytes", C.c_uint32), ("mtime", C.c_uint32),
        ("d_out", C.c_void_p), ("out_cap", C.c_uint64), ("d_out_bytes", C.c_void_p),
    ]


class _GzBatchC(C.Structure):
    _fields_ = [
        ("d_src", C.c_void_p), ("src_bytes", C.c_uint64),
        ("d_members", C.c_void_p), ("n_members", C.c_uint32),
        ("d_dst", C.c_void_p), ("dst_cap", C.c_uint64),
        ("d_results", C.c_void_p), ("d_summary", C.c_void_p),
        ("options", C.c_uint32), ("reserved", C.c_uint32),
    ]


class _Lz4IndexC(C.Structure):
    _fields_ = [
        ("blocks", C.c_void_p), ("n_blocks", C.c_uint32), ("cap_blocks", C.c_uint32),
        ("frames", C.c_void_p), ("n_frames", C.c_uint32), ("cap_frames", C.c_uint32),
        ("end_kind", C.c_int), ("consumed", C.c_uint64), ("max_out", C.c_uint64),
    ]


_gpu = None
_host = None


def gpu_lib():
    """libla_gpu.so (HIP kernels + C shim).  Raises when it has not been built."""
    global _gpu
    if _gpu is None:
        if not os.path.exists(GPU_LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found: build it with `make -C libarchive_amd/csrc` (hipcc, gfx950); "
                "there is no CPU fallback" % GPU_LIB_PATH)
        # This harness shares device buffers with PyTorch, and the PyTorch wheel carries its own HIP runtime: it has to
        # be the one already loaded when libla_gpu.so resolves libamdhip64, or the process ends up with two runtimes and
        # la_gpu_open() sees no device (found with `python __graft_entry__.py smoke`, where build() loaded this library
        # before smoke() imported torch).  A C caller (la_cat, the filters inside libarchive) links the system runtime
        # and never meets torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(GPU_LIB_PATH)
        lib.la_gpu_abi_version.restype = C.c_int
        lib.la_gpu_device_count.restype = C.c_int
        lib.la_gpu_open.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.la_gpu_close.argtypes = [C.c_void_p]
        lib.la_gpu_close.restype = None
        lib.la_gpu_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.la_gpu_sync.argtypes = [C.c_void_p]
        lib.la_gpu_last_error.argtypes = [C.c_void_p]
        lib.la_gpu_last_error.restype = C.c_char_p
        lib.la_gpu_reserve.argtypes = [C.c_void_p, C.c_uint64]
        lib.la_gpu_timer_start.argtypes = [C.c_void_p]
        lib.la_gpu_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.la_gpu_profile_enable.argtypes = [C.c_void_p, C.c_int]
        lib.la_gpu_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.c_int]
        lib.la_gpu_xxh32_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.la_gpu_crc32_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.la_gpu_lz4_workspace_bytes.argtypes = [C.c_uint32, C.c_uint64]
        lib.la_gpu_lz4_workspace_bytes.restype = C.c_uint64
        lib.la_gpu_lz4_decode.argtypes = [C.c_void_p, C.POINTER(_Lz4BatchC)]
        lib.la_gpu_lz4_compress.argtypes = [C.c_void_p, C.POINTER(_Lz4cBatchC)]
        lib.la_gpu_gzip_compress.argtypes = [C.c_void_p, C.POINTER(_GzcBatchC)]
        lib.la_gpu_gzip_compress_bound.restype = C.c_uint64
        lib.la_gpu_gzip_compress_bound.argtypes = [C.c_uint64, C.c_uint32]
        lib.la_gpu_gzip_compress_workspace_bytes.restype = C.c_uint64
        lib.la_gpu_gzip_compress_workspace_bytes.argtypes = [C.c_uint64, C.c_uint32]
        lib.la_gpu_lz4_compress_bound.restype = C.c_uint64
        lib.la_gpu_lz4_compress_bound.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        lib.la_gpu_lz4_compress_workspace_bytes.restype = C.c_uint64
        lib.la_gpu_lz4_compress_workspace_bytes.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        lib.la_gpu_gzip_decode.argtypes = [C.c_void_p, C.POINTER(_GzBatchC)]
        _gpu = lib
    return _gpu


def host_lib():
    """libla_host.so (plain-C walkers / read core / filters)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found: build it with `make -C libarchive_amd/host`" % HOST_LIB_PATH)
        lib = C.CDLL(HOST_LIB_PATH)
        lib.la_lz4_index_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(_Lz4IndexC)]
        lib.la_lz4_index_free.argtypes = [C.POINTER(_Lz4IndexC)]
        lib.la_lz4_index_free.restype = None
        lib.la_lz4_bid_bytes.argtypes = [C.c_char_p, C.c_size_t]
        lib.la_status_message.argtypes = [C.c_uint32]
        lib.la_status_message.restype = C.c_char_p
        lib.la_end_message.argtypes = [C.c_int, C.c_int]
        lib.la_end_message.restype = C.c_char_p
        _host = lib
    return _host


def status_message(st: int) -> str:
    return host_lib().la_status_message(int(st)).decode()


def end_message(kind: int, is_gzip: bool = False) -> str:
    return host_lib().la_end_message(int(kind), 1 if is_gzip else 0).decode()


class Lz4Index:
    """Block / frame job tables of a .lz4 image (host walker, la_lz4_index_build)."""

    def __init__(self, blocks, frames, end_kind, consumed, max_out):
        self.blocks = blocks
        self.frames = frames
        self.end_kind = end_kind
        self.consumed = consumed
        self.max_out = max_out


def lz4_index(image, at_eof: bool = True) -> Lz4Index:
    """Walk a compressed image (bytes or uint8 ndarray) on the host."""
    if isinstance(image, (bytes, bytearray, memoryview)):
        image = np.frombuffer(bytes(image), dtype=np.uint8)
    image = np.ascontiguousarray(image, dtype=np.uint8)
    c = _Lz4IndexC()
    rc = host_lib().la_lz4_index_build(image.ctypes.data, image.size, 1 if at_eof else 0, C.byref(c))
    if rc != 0:
        raise MemoryError("la_lz4_index_build")
    try:
        blocks = np.empty(c.n_blocks, dtype=LZ4_BLOCK_DTYPE)
        frames = np.empty(c.n_frames, dtype=LZ4_FRAME_DTYPE)
        if c.n_blocks:
            C.memmove(blocks.ctypes.data, c.blocks, blocks.nbytes)
        if c.n_frames:
            C.memmove(frames.ctypes.data, c.frames, frames.nbytes)
        return Lz4Index(blocks, frames, c.end_kind, c.consumed, c.max_out)
    finally:
        host_lib().la_lz4_index_free(C.byref(c))


GZ_HEADER_DTYPE = np.dtype([("off", "<u8"), ("len", "<u4"), ("mtime", "<u4"), ("name_off", "<u4"), ("bgzf_size", "<u4")])


class _GzIndexC(C.Structure):
    _fields_ = [("members", C.c_void_p), ("headers", C.c_void_p), ("n", C.c_uint32), ("cap", C.c_uint32),
                ("end_kind", C.c_int), ("consumed", C.c_uint64), ("max_out", C.c_uint64), ("speculative", C.c_int)]


class GzIndex:
    """Member table of a .gz image (host walker, la_gz_index_build)."""

    def __init__(self, members, headers, end_kind, consumed, max_out, speculative):
        self.members, self.headers = members, headers
        self.end_kind, self.consumed, self.max_out, self.speculative = end_kind, consumed, max_out, speculative


def gz_index(image, at_eof: bool = True) -> GzIndex:
    if isinstance(image, (bytes, bytearray, memoryview)):
        image = np.frombuffer(bytes(image), dtype=np.uint8)
    image = np.ascontiguousarray(image, dtype=np.uint8)
    lib = host_lib()
    lib.la_gz_index_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(_GzIndexC)]
    lib.la_gz_index_free.argtypes = [C.POINTER(_GzIndexC)]
    lib.la_gz_index_free.restype = None
    c = _GzIndexC()
    if lib.la_gz_index_build(image.ctypes.data, image.size, 1 if at_eof else 0, C.byref(c)) != 0:
        raise MemoryError("la_gz_index_build")
    try:
        mem = np.empty(c.n, dtype=GZ_MEMBER_DTYPE)
        hdr = np.empty(c.n, dtype=GZ_HEADER_DTYPE)
        if c.n:
            C.memmove(mem.ctypes.data, c.members, mem.nbytes)
            C.memmove(hdr.ctypes.data, c.headers, hdr.nbytes)
        return GzIndex(mem, hdr, c.end_kind, c.consumed, c.max_out, c.speculative)
    finally:
        lib.la_gz_index_free(C.byref(c))


class GpuContext:
    """la_gpu_ctx: one HIP stream + workspace on one device."""

    def __init__(self, device: int = 0, stream=None):
        lib = gpu_lib()
        h = C.c_void_p()
        rc = lib.la_gpu_open(device, C.byref(h))
        if rc != LA_OK:
            raise RuntimeError("la_gpu_open(%d) failed with %d: no usable gfx950 device "
                               "(there is no CPU fallback)" % (device, rc))
        self._h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def _check(self, rc, what):
        if rc != LA_OK:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, gpu_lib().la_gpu_last_error(self._h).decode()))

    def set_stream(self, cuda_stream_handle):
        self._check(gpu_lib().la_gpu_set_stream(self._h, C.c_void_p(cuda_stream_handle)), "la_gpu_set_stream")

    def reserve(self, nbytes):
        self._check(gpu_lib().la_gpu_reserve(self._h, int(nbytes)), "la_gpu_reserve")

    def sync(self):
        self._check(gpu_lib().la_gpu_sync(self._h), "la_gpu_sync")

    def timer_start(self):
        self._check(gpu_lib().la_gpu_timer_start(self._h), "la_gpu_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(gpu_lib().la_gpu_timer_stop(self._h, C.byref(ms)), "la_gpu_timer_stop")
        return ms.value

    def profile_enable(self, on=True):
        self._check(gpu_lib().la_gpu_profile_enable(self._h, 1 if on else 0), "la_gpu_profile_enable")

    def profile_read(self):
        """{phase name: ms} of the most recent batch call (synchronises)."""
        ms = (C.c_float * 12)()
        names = (C.c_char_p * 12)()
        n = gpu_lib().la_gpu_profile_read(self._h, ms, names, 12)
        if n < 0:
            self._check(n, "la_gpu_profile_read")
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def xxh32_many(self, d_base_ptr, d_jobs_ptr, n, d_out_ptr):
        self._check(gpu_lib().la_gpu_xxh32_many(self._h, d_base_ptr, d_jobs_ptr, n, d_out_ptr), "la_gpu_xxh32_many")

    def crc32_many(self, d_base_ptr, d_jobs_ptr, n, d_out_ptr):
        self._check(gpu_lib().la_gpu_crc32_many(self._h, d_base_ptr, d_jobs_ptr, n, d_out_ptr), "la_gpu_crc32_many")

    def lz4_decode(self, batch: _Lz4BatchC):
        self._check(gpu_lib().la_gpu_lz4_decode(self._h, C.byref(batch)), "la_gpu_lz4_decode")

    def lz4_compress(self, batch: _Lz4cBatchC):
        self._check(gpu_lib().la_gpu_lz4_compress(self._h, C.byref(batch)), "la_gpu_lz4_compress")

    def gzip_compress(self, batch: _GzcBatchC):
        self._check(gpu_lib().la_gpu_gzip_compress(self._h, C.byref(batch)), "la_gpu_gzip_compress")

    def gzip_decode(self, batch: _GzBatchC):
        self._check(gpu_lib().la_gpu_gzip_decode(self._h, C.byref(batch)), "la_gpu_gzip_decode")

    def close(self):
        if getattr(self, "_h", None):
            gpu_lib().la_gpu_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
