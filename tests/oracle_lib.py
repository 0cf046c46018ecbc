"""ctypes binding of the CPU oracle (oracle/liblaoracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (libarchive_amd) never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "liblaoracle.so")

ARCHIVE_FATAL = -30


class StreamResult(C.Structure):
    _fields_ = [
        ("rc", C.c_int),
        ("errmsg", C.c_char * 96),
        ("out_len", C.c_size_t),
        ("gz_mtime", C.c_uint32),
        ("gz_name", C.c_char * 256),
        ("gz_has_name", C.c_int),
        ("n_units", C.c_uint64),
        ("n_frames", C.c_uint64),
        ("gz_trailer_mismatch", C.c_int),
    ]


class XXHState(C.Structure):
    _fields_ = [
        ("total_len", C.c_uint64),
        ("seed", C.c_uint32),
        ("v", C.c_uint32 * 4),
        ("memsize", C.c_uint32),
        ("mem", C.c_uint8 * 16),
    ]


def build():
    """(Re)build the oracle with the committed Makefile (gcc only)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    lib.orc_xxh32.restype = C.c_uint32
    lib.orc_xxh32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    lib.orc_xxh32_init.argtypes = [C.POINTER(XXHState), C.c_uint32]
    lib.orc_xxh32_update.argtypes = [C.POINTER(XXHState), C.c_char_p, C.c_size_t]
    lib.orc_xxh32_digest.restype = C.c_uint32
    lib.orc_xxh32_digest.argtypes = [C.POINTER(XXHState)]
    lib.orc_crc32.restype = C.c_uint32
    lib.orc_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
    lib.orc_crc32_combine.restype = C.c_uint32
    lib.orc_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
    lib.orc_lz4_block_decode.restype = C.c_int
    lib.orc_lz4_block_decode.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    lib.orc_inflate_raw.restype = C.c_int
    lib.orc_inflate_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    for f in (lib.orc_lz4_stream_decode, lib.orc_gzip_stream_decode):
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(StreamResult)]
    lib.orc_lz4_bid.restype = C.c_int
    lib.orc_lz4_bid.argtypes = [C.c_char_p, C.c_size_t]
    lib.orc_gzip_bid.restype = C.c_int
    lib.orc_gzip_bid.argtypes = [C.c_char_p, C.c_size_t]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def xxh32(data: bytes, seed: int = 0) -> int:
    return lib().orc_xxh32(data, len(data), seed)


def xxh32_stream(chunks, seed: int = 0) -> int:
    st = XXHState()
    lib().orc_xxh32_init(C.byref(st), seed)
    for c in chunks:
        lib().orc_xxh32_update(C.byref(st), c, len(c))
    return lib().orc_xxh32_digest(C.byref(st))


def crc32(data: bytes, crc: int = 0) -> int:
    return lib().orc_crc32(crc, data, len(data))


def crc32_combine(a: int, b: int, len_b: int) -> int:
    return lib().orc_crc32_combine(a, b, len_b)


def lz4_block_decode(src: bytes, dst_cap: int, dict_: bytes = b""):
    """Returns decoded bytes, or None when the block is rejected."""
    buf = C.create_string_buffer(max(dst_cap, 1))
    n = lib().orc_lz4_block_decode(src, len(src), buf, dst_cap,
                                   dict_ if dict_ else None, len(dict_))
    if n < 0:
        return None
    return buf.raw[:n]


def inflate_raw(src: bytes, dst_cap: int):
    """Returns (rc, consumed, output_bytes)."""
    buf = C.create_string_buffer(max(dst_cap, 1))
    cons = C.c_size_t(0)
    prod = C.c_size_t(0)
    rc = lib().orc_inflate_raw(src, len(src), buf, dst_cap, C.byref(cons), C.byref(prod))
    return rc, cons.value, buf.raw[:prod.value]


def _stream(fn, src, out_cap):
    import numpy as np
    res = StreamResult()
    if isinstance(src, (bytes, bytearray)):
        src = np.frombuffer(bytes(src), dtype=np.uint8)
    src = np.ascontiguousarray(src)
    out = np.empty(max(out_cap, 1), dtype=np.uint8)
    r = fn(src.ctypes.data, src.size, out.ctypes.data, out_cap, C.byref(res))
    if r != 0:
        raise RuntimeError("oracle output buffer too small")
    return out[:res.out_len], res


def lz4_stream_decode(src, out_cap: int):
    """(numpy uint8 output, StreamResult) for a whole .lz4 image."""
    return _stream(lib().orc_lz4_stream_decode, src, out_cap)


def gzip_stream_decode(src, out_cap: int):
    return _stream(lib().orc_gzip_stream_decode, src, out_cap)


def lz4_bid(b: bytes) -> int:
    return lib().orc_lz4_bid(b, len(b))


def gzip_bid(b: bytes) -> int:
    return lib().orc_gzip_bid(b, len(b))


# ---- the REAL reference hash code (oracle/_ref/libref_hash.so), when built ----

class _RefXX(C.Structure):
    _fields_ = [("XXH32", C.c_void_p), ("init", C.c_void_p), ("update", C.c_void_p), ("digest", C.c_void_p)]


def ref_hash():
    """Returns (ref_xxh32(data, seed), ref_xxh32_stream(chunks, seed), ref_crc32(data, crc)) or None."""
    path = os.path.join(ORACLE_DIR, "_ref", "libref_hash.so")
    if not os.path.exists(path):
        return None
    r = C.CDLL(path)
    tab = _RefXX.in_dll(r, "__archive_xxhash")
    f_xxh = C.CFUNCTYPE(C.c_uint, C.c_char_p, C.c_uint, C.c_uint)(tab.XXH32)
    f_init = C.CFUNCTYPE(C.c_void_p, C.c_uint)(tab.init)
    f_upd = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_uint)(tab.update)
    f_dig = C.CFUNCTYPE(C.c_uint, C.c_void_p)(tab.digest)
    r.ref_crc32.restype = C.c_ulong
    r.ref_crc32.argtypes = [C.c_ulong, C.c_char_p, C.c_size_t]

    def rx(data, seed=0):
        return f_xxh(data, len(data), seed)

    def rxs(chunks, seed=0):
        st = f_init(seed)
        for c in chunks:
            f_upd(st, c, len(c))
        return f_dig(st)  # frees the state (xxhash.c:504)

    def rc(data, crc=0):
        return r.ref_crc32(crc, data, len(data)) & 0xFFFFFFFF

    return rx, rxs, rc
