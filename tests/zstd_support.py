"""Shared helpers of the zstd tests: the image's libzstd through ctypes (compressor for test inputs and the library the
reference's filter calls, archive_read_support_filter_zstd.c:226), the oracle's stream decoder, input generators."""
import ctypes
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def libzstd():
    try:
        z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return None
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    z.ZSTD_isError.restype = ctypes.c_uint
    z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    z.ZSTD_versionNumber.restype = ctypes.c_uint
    return z


def zstd_compress(z, data, level):
    cap = z.ZSTD_compressBound(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress(buf, cap, data, len(data), level)
    assert not z.ZSTD_isError(n)
    return buf.raw[:n]


def zstd_decompress(z, img, cap):
    buf = ctypes.create_string_buffer(max(cap, 1))
    n = z.ZSTD_decompress(buf, cap, img, len(img))
    return None if z.ZSTD_isError(n) else buf.raw[:n]


def oracle_lib():
    o = ctypes.CDLL(os.path.join(ROOT, "oracle", "liblaoracle.so"))
    o.orc_zstd_stream_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.c_size_t]
    o.orc_zstd_bid.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    o.orc_xxh64.restype = ctypes.c_uint64
    o.orc_xxh64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
    return o


def oracle_decode(o, img, cap):
    buf = ctypes.create_string_buffer(max(cap, 1))
    n = ctypes.c_size_t(0)
    msg = ctypes.create_string_buffer(128)
    rc = o.orc_zstd_stream_decode(img, len(img), buf, cap, ctypes.byref(n), msg, 128)
    return rc, buf.raw[:n.value], msg.value.decode()


def gen(rnd, n, kind):
    """random bytes / small alphabet / words / one byte / LZ-shaped copies"""
    if kind == 0:
        return rnd.randbytes(n) if hasattr(rnd, "randbytes") else bytes(rnd.getrandbits(8) for _ in range(n))
    if kind == 1:
        return bytes(rnd.choice(b"abcdefgh ") for _ in range(n))
    if kind == 2:
        w = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(2, 9))) for _ in range(50)]
        out = bytearray()
        while len(out) < n:
            out += rnd.choice(w) + b" "
        return bytes(out[:n])
    if kind == 3:
        return bytes([rnd.randint(0, 255)]) * n
    out = bytearray()
    while len(out) < n:
        if out and rnd.random() < 0.5:
            off = rnd.randint(1, len(out))
            for _ in range(rnd.randint(3, 300)):
                out.append(out[-off])
        else:
            out += bytes(rnd.getrandbits(8) for _ in range(rnd.randint(1, 40)))
    return bytes(out[:n])


def skippable(payload, nibble=0):
    return (0x184D2A50 + nibble).to_bytes(4, "little") + len(payload).to_bytes(4, "little") + payload
