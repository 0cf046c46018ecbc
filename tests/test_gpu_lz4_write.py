"""The lz4 WRITE filter on the device data plane through the archive_write_* slice
(host/la_write_filters.c): archive_write_new -> add_filter_lz4 -> set_format_raw -> open_memory -> header ->
data (in pieces) -> close.  What it writes must read back as the input through the reference-equivalent
reader (the oracle), through this repository's own read path (la_api.cat: bid, filter, raw format), and must
carry the options in its frame descriptors."""
import ctypes as C
import random

import pytest

import la_api
import oracle_lib as O
import libarchive_amd as la

pytestmark = pytest.mark.gpu
ARCHIVE_OK, ARCHIVE_FAILED, ARCHIVE_FATAL = 0, -25, -30


def _lib():
    return _lib_setup(la.host_lib())


def _lib_setup(lib):
    lib.archive_write_new.restype = C.c_void_p
    for f in ("archive_write_add_filter_lz4", "archive_write_add_filter_gzip", "archive_write_set_format_raw", "archive_write_close", "archive_write_free"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.archive_write_set_filter_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.archive_write_open_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.archive_write_header.argtypes = [C.c_void_p, C.c_void_p]
    lib.archive_write_data.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.archive_write_data.restype = C.c_ssize_t
    lib.archive_error_string.argtypes = [C.c_void_p]
    lib.archive_error_string.restype = C.c_char_p
    return lib


def write_lz4(data, options=(), piece=None, cap=None, codec="lz4"):
    lib = _lib()
    a = lib.archive_write_new()
    assert getattr(lib, "archive_write_add_filter_" + codec)(a) == ARCHIVE_OK
    assert lib.archive_write_set_format_raw(a) == ARCHIVE_OK
    for k, v in options:
        rc = lib.archive_write_set_filter_option(a, codec.encode(), k.encode(), None if v is None else v.encode())
        if rc != ARCHIVE_OK:
            err = lib.archive_error_string(a)
            lib.archive_write_free(a)
            return rc, err.decode() if err else None
    cap = cap if cap is not None else len(data) + len(data) // 200 + 65536
    buf = C.create_string_buffer(cap)
    used = C.c_size_t(0)
    assert lib.archive_write_open_memory(a, buf, cap, C.byref(used)) == ARCHIVE_OK
    assert lib.archive_write_header(a, None) == ARCHIVE_OK
    step = piece or max(len(data), 1)
    for i in range(0, len(data), step):
        chunk = data[i:i + step]
        r = lib.archive_write_data(a, chunk, len(chunk))
        if r != len(chunk):
            err = lib.archive_error_string(a)
            lib.archive_write_free(a)
            return r, err.decode() if err else None
    rc = lib.archive_write_close(a)
    out = buf.raw[:used.value]
    if rc != ARCHIVE_OK:
        err = lib.archive_error_string(a)
        out = err.decode() if err else None
    lib.archive_write_free(a)
    return rc, out


def test_write_filter_round_trips(gpu_ctx, monkeypatch):
    monkeypatch.setenv("LA_GPU_WRITE_WINDOW_MIB", "2")      # several windows for a few MiB of input
    rnd = random.Random(9)
    words = [rnd.randbytes(rnd.randint(2, 10)) for _ in range(200)]
    text = b"".join(rnd.choice(words) for _ in range(1200000))[:5 * 1024 * 1024 + 333]
    for data, piece in ((b"", None), (b"x", None), (text[:70000], 1000), (text, 65536 + 17), (rnd.randbytes(300000), None), (bytes(3 << 20), 4096)):
        for options in ((), (("block-checksum", "1"),), (("stream-checksum", None), ("block-size", "4"), ("compression-level", "3"))):
            rc, img = write_lz4(data, options, piece)
            assert rc == ARCHIVE_OK and isinstance(img, bytes)
            assert img[:4] == b"\x04\x22\x4d\x18"
            flg = img[4]
            assert (flg & 0xC0) == 0x40 and (flg & 0x20)                      # version 01, independent blocks
            assert bool(flg & 0x10) == (("block-checksum", "1") in options)
            assert bool(flg & 0x04) == (("stream-checksum", None) not in options)
            out, res = O.lz4_stream_decode(img, len(data) + 64)               # the reference reader, restated
            assert (res.rc, res.errmsg) == (0, b"") and out.tobytes() == data
            r = la_api.cat(img)                                               # this repository's read path
            assert r.filters[0] == (13, "lz4") and r.data == data and r.rc == la_api.ARCHIVE_EOF
            if len(data) > 100000 and data[:16] != rnd.randbytes(0):
                assert len(img) < len(data) + 1024


def test_write_filter_options_and_errors(gpu_ctx):
    rc, err = write_lz4(b"abc", (("block-dependence", "1"),))
    assert rc == ARCHIVE_FAILED and "block dependence" in err
    rc, err = write_lz4(b"abc", (("no-such-option", "1"),))
    assert rc == ARCHIVE_FAILED and "Undefined option" in err
    rc, err = write_lz4(bytes(200000), (), None, cap=100)     # the client's buffer is too small
    assert rc == ARCHIVE_FATAL and err == "Buffer exhausted"


def test_gzip_write_filter_round_trips(gpu_ctx, monkeypatch):
    """archive_write_add_filter_gzip on the device data plane: what it writes reads back through zlib's gzip reader,
    the oracle's gzip filter and this repository's read path; options as the reference's (timestamp, compression-level)."""
    import gzip
    import io
    import struct
    monkeypatch.setenv("LA_GPU_WRITE_WINDOW_MIB", "2")
    rnd = random.Random(10)
    words = [rnd.randbytes(rnd.randint(2, 10)) for _ in range(200)]
    text = b"".join(rnd.choice(words) for _ in range(1200000))[:5 * 1024 * 1024 + 77]
    for data, piece in ((b"", None), (b"z", None), (text[:50000], 999), (text, 65536 + 3), (rnd.randbytes(200000), None), (bytes(2 << 20), 8192)):
        for options in ((), (("timestamp", None),), (("compression-level", "9"),)):
            rc, img = write_lz4(data, options, piece, codec="gzip")
            assert rc == ARCHIVE_OK and img[:3] == b"\x1f\x8b\x08"
            mtime = struct.unpack_from("<I", img, 4)[0]
            assert (mtime == 0) == (("timestamp", None) in options)
            assert gzip.GzipFile(fileobj=io.BytesIO(img)).read() == data
            out, res = O.gzip_stream_decode(img, len(data) + 64)
            assert (res.rc, res.errmsg) == (0, b"") and out.tobytes() == data
            r = la_api.cat(img)
            assert r.filters[0] == (1, "gzip") and r.data == data
    rc, err = write_lz4(b"abc", (("no-such-option", "1"),), codec="gzip")
    assert rc == ARCHIVE_FAILED and "Undefined option" in err
