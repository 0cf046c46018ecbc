"""ctypes harness over the public archive_read_* slice exported by libla_host.so
(include/la_archive.h).  Mirrors how bsdcat drives libarchive (cat/bsdcat.c:74-94)."""
import ctypes as C
import os

import libarchive_amd as la

ARCHIVE_EOF, ARCHIVE_OK, ARCHIVE_FATAL = 1, 0, -30


_OVERRIDE = None   # a ctypes CDLL to use instead of the product's libla_host.so (tests/mock_gpu)


def use_library(cdll):
    """Route this harness to another build of the host library (None = the product's)."""
    global _OVERRIDE
    _OVERRIDE = cdll


def _lib():
    lib = _OVERRIDE if _OVERRIDE is not None else la.host_lib()
    if getattr(lib, "_la_api_ready", False):
        return lib
    lib.archive_read_new.restype = C.c_void_p
    for f in ("archive_read_support_filter_all", "archive_read_support_filter_gzip", "archive_read_support_filter_lz4",
              "archive_read_support_format_raw", "archive_read_support_format_empty", "archive_read_close",
              "archive_read_free", "archive_filter_count", "archive_errno", "archive_format",
              "archive_read_support_format_tar", "archive_read_support_format_all", "archive_read_data_skip"):
        getattr(lib, f).argtypes = [C.c_void_p]
        getattr(lib, f).restype = C.c_int
    lib.archive_read_open_memory2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    lib.archive_read_open_filename.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.archive_read_next_header.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.archive_read_data_block.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    lib.archive_read_data.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.archive_read_data.restype = C.c_ssize_t
    lib.archive_error_string.argtypes = [C.c_void_p]
    lib.archive_error_string.restype = C.c_char_p
    lib.archive_filter_code.argtypes = [C.c_void_p, C.c_int]
    lib.archive_filter_name.argtypes = [C.c_void_p, C.c_int]
    lib.archive_filter_name.restype = C.c_char_p
    lib.archive_filter_bytes.argtypes = [C.c_void_p, C.c_int]
    lib.archive_filter_bytes.restype = C.c_int64
    lib.archive_format_name.argtypes = [C.c_void_p]
    lib.archive_format_name.restype = C.c_char_p
    lib.archive_entry_pathname.argtypes = [C.c_void_p]
    lib.archive_entry_pathname.restype = C.c_char_p
    lib.archive_entry_mtime.argtypes = [C.c_void_p]
    lib.archive_entry_mtime.restype = C.c_int64
    lib.archive_entry_mtime_is_set.argtypes = [C.c_void_p]
    lib.archive_entry_size.argtypes = [C.c_void_p]
    lib.archive_entry_size.restype = C.c_int64
    lib.archive_entry_filetype.argtypes = [C.c_void_p]
    lib.archive_entry_filetype.restype = C.c_uint
    lib.archive_entry_perm.argtypes = [C.c_void_p]
    lib.archive_entry_perm.restype = C.c_uint
    lib._la_api_ready = True
    return lib


class CatResult:
    def __init__(self):
        self.data = b""
        self.open_rc = None
        self.rc = None          # final return code of the read loop (ARCHIVE_EOF on a clean end)
        self.error = None       # archive_error_string or None
        self.filters = []       # [(code, name)] head first
        self.pathname = None
        self.mtime = None
        self.block_sizes = []
        self.format_name = None
        self.bytes_in = None
        self.bytes_out = None


def cat(image, read_size=None, use_read_data=None, filename=None, block_size=10240):
    """bsdcat-equivalent: support_filter_all + format_empty + format_raw, then the data-block loop."""
    lib = _lib()
    a = lib.archive_read_new()
    res = CatResult()
    try:
        lib.archive_read_support_filter_all(a)
        lib.archive_read_support_format_empty(a)
        lib.archive_read_support_format_raw(a)
        if filename is not None:
            res.open_rc = lib.archive_read_open_filename(a, filename.encode(), block_size)
        else:
            buf = C.create_string_buffer(bytes(image), len(image))
            res._keep = buf
            res.open_rc = lib.archive_read_open_memory2(a, buf, len(image), read_size or max(len(image), 1))
        if res.open_rc != ARCHIVE_OK:
            e = lib.archive_error_string(a)
            res.error = e.decode() if e else None
            res.rc = res.open_rc
            return res
        n = lib.archive_filter_count(a)
        res.filters = [(lib.archive_filter_code(a, i), lib.archive_filter_name(a, i).decode()) for i in range(n)]
        ent = C.c_void_p()
        r = lib.archive_read_next_header(a, C.byref(ent))
        if r == ARCHIVE_OK:
            res.pathname = lib.archive_entry_pathname(ent).decode()
            res.mtime = lib.archive_entry_mtime(ent) if lib.archive_entry_mtime_is_set(ent) else None
            out = bytearray()
            if use_read_data:
                tmp = C.create_string_buffer(use_read_data)
                while True:
                    k = lib.archive_read_data(a, tmp, use_read_data)
                    if k <= 0:
                        r = ARCHIVE_EOF if k == 0 else int(k)
                        break
                    out += tmp.raw[:k]
            else:
                p, sz, off = C.c_void_p(), C.c_size_t(), C.c_int64()
                while True:
                    r = lib.archive_read_data_block(a, C.byref(p), C.byref(sz), C.byref(off))
                    if r != ARCHIVE_OK:
                        break
                    assert off.value == len(out)
                    res.block_sizes.append(sz.value)
                    out += C.string_at(p.value, sz.value)
            res.data = bytes(out)
        res.rc = r
        e = lib.archive_error_string(a)
        res.error = e.decode() if e else None
        fn = lib.archive_format_name(a)
        res.format_name = fn.decode() if fn else None
        res.bytes_in = lib.archive_filter_bytes(a, -1)
        res.bytes_out = lib.archive_filter_bytes(a, 0)
        return res
    finally:
        lib.archive_read_free(a)


def as_reference_tuple(res):
    """(bytes, rc, message) in the oracle's convention: rc 0 for a clean end, -30 fatal."""
    if res.rc in (ARCHIVE_EOF, ARCHIVE_OK):
        return res.data, 0, ""
    return res.data, ARCHIVE_FATAL, res.error or ""


class ListResult:
    def __init__(self):
        self.open_rc = None
        self.entries = []       # [(pathname, size, filetype, perm, mtime, data or None)]
        self.rc = None          # what ended the header loop (ARCHIVE_EOF on a clean end)
        self.error = None
        self.filters = []
        self.format = None
        self.format_name = None


def list_entries(image, read_size=None, filename=None, block_size=10240, read_bodies=True, skip_every=0):
    """bsdtar -t / -x shape: support_filter_all + support_format_all, then the
    archive_read_next_header / archive_read_data_block loop (tar/read.c:206-400).  With
    `skip_every` = k every k-th body is left unread so that next_header has to skip it."""
    lib = _lib()
    a = lib.archive_read_new()
    res = ListResult()
    try:
        lib.archive_read_support_filter_all(a)
        lib.archive_read_support_format_all(a)
        if filename is not None:
            res.open_rc = lib.archive_read_open_filename(a, filename.encode(), block_size)
        else:
            buf = C.create_string_buffer(bytes(image), len(image))
            res._keep = buf
            res.open_rc = lib.archive_read_open_memory2(a, buf, len(image), read_size or max(len(image), 1))
        if res.open_rc != ARCHIVE_OK:
            e = lib.archive_error_string(a)
            res.error = e.decode() if e else None
            res.rc = res.open_rc
            return res
        n = lib.archive_filter_count(a)
        res.filters = [(lib.archive_filter_code(a, i), lib.archive_filter_name(a, i).decode()) for i in range(n)]
        ent = C.c_void_p()
        p, sz, off = C.c_void_p(), C.c_size_t(), C.c_int64()
        i = 0
        while True:
            r = lib.archive_read_next_header(a, C.byref(ent))
            if r != ARCHIVE_OK:
                break
            i += 1
            meta = (lib.archive_entry_pathname(ent).decode("latin-1"), lib.archive_entry_size(ent),
                    lib.archive_entry_filetype(ent), lib.archive_entry_perm(ent), lib.archive_entry_mtime(ent))
            body = None
            if read_bodies and not (skip_every and i % skip_every == 0):
                out = bytearray()
                while True:
                    r = lib.archive_read_data_block(a, C.byref(p), C.byref(sz), C.byref(off))
                    if r != ARCHIVE_OK:
                        break
                    assert off.value == len(out)
                    out += C.string_at(p.value, sz.value)
                body = bytes(out)
                if r != ARCHIVE_EOF:
                    res.entries.append(meta + (body,))
                    break
            res.entries.append(meta + (body,))
        res.rc = r
        e = lib.archive_error_string(a)
        res.error = e.decode() if e else None
        res.format = lib.archive_format(a)
        fn = lib.archive_format_name(a)
        res.format_name = fn.decode() if fn else None
        return res
    finally:
        lib.archive_read_free(a)
