"""libarchive_amd -- MI355X-native data plane for libarchive's lz4 / gzip read filters.

The product is native: HIP kernels + an extern "C" shim (csrc/ -> libla_gpu.so,
ABI in include/la_gpu.h) and plain-C host code (host/ -> libla_host.so, ABI in
include/la_host.h, include/la_filter.h).  This Python package is only a thin
ctypes harness over that C ABI for tests and bench.py; PyTorch is used for
device buffers, streams and torch.distributed -- plumbing, not product.

There is NO CPU fallback: loading fails loudly when the native libraries are
missing, and every decode call fails when no gfx950 device is present.
"""
from ._native import (  # noqa: F401
    GpuContext,
    NativeLibraryMissing,
    LZ4_BLOCK_DTYPE,
    LZ4_FRAME_DTYPE,
    HASH_JOB_DTYPE,
    SUMMARY_DTYPE,
    Lz4Index,
    gpu_lib,
    host_lib,
    lz4_index,
    gz_index,
    GzIndex,
    GZ_MEMBER_DTYPE,
    GZ_RESULT_DTYPE,
    status_message,
    end_message,
)
from . import lz4  # noqa: F401

__all__ = [
    "GpuContext", "NativeLibraryMissing", "Lz4Index", "lz4_index", "gpu_lib", "host_lib",
    "LZ4_BLOCK_DTYPE", "LZ4_FRAME_DTYPE", "HASH_JOB_DTYPE", "SUMMARY_DTYPE", "lz4",
    "status_message", "end_message",
]
