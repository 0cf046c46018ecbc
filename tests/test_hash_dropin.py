"""The host-callable hash drop-ins (host/la_hash_dropin.c): the `__archive_xxhash`-shaped table and
crc32() with zlib's signature, against the REAL reference code compiled into oracle/_ref (when the
reference tree is present), the oracle, and zlib.  The large-buffer crc32 path (device batches +
GF(2) combine on the host) has its own `gpu` test."""
import ctypes as C
import random
import zlib

import pytest

import oracle_lib as O
import libarchive_amd as la


class _Tab(C.Structure):
    _fields_ = [("XXH32", C.c_void_p), ("init", C.c_void_p), ("update", C.c_void_p), ("digest", C.c_void_p)]


def _table():
    lib = la.host_lib()
    tab = _Tab.in_dll(lib, "la_archive_xxhash")
    f_xxh = C.CFUNCTYPE(C.c_uint, C.c_char_p, C.c_uint, C.c_uint)(tab.XXH32)
    f_init = C.CFUNCTYPE(C.c_void_p, C.c_uint)(tab.init)
    f_upd = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_uint)(tab.update)
    f_dig = C.CFUNCTYPE(C.c_uint, C.c_void_p)(tab.digest)
    lib.la_crc32.restype = C.c_ulong
    lib.la_crc32.argtypes = [C.c_ulong, C.c_char_p, C.c_size_t]
    lib.la_crc32_host.restype = C.c_ulong
    lib.la_crc32_host.argtypes = [C.c_ulong, C.c_char_p, C.c_size_t]
    return lib, f_xxh, f_init, f_upd, f_dig


def test_xxhash_table_and_crc32_against_the_reference_code():
    lib, f_xxh, f_init, f_upd, f_dig = _table()
    ref = O.ref_hash()      # real libarchive/xxhash.c + archive_crc32.h, or None on a box without the tree
    rnd = random.Random(21)
    for n in list(range(0, 70)) + [255, 256, 1000, 4095, 65536, 100003]:
        d = rnd.randbytes(n)
        s = rnd.getrandbits(32)
        want = O.xxh32(d, s)
        if ref:
            assert ref[0](d, s) == want
        assert f_xxh(d, n, s) == want
        # streaming: every split into up to four pieces hashes like the whole (state is malloc'ed, digest frees it)
        cuts = sorted(rnd.randint(0, n) for _ in range(3))
        pieces = [d[a:b] for a, b in zip([0] + cuts, cuts + [n])]
        st = f_init(s)
        assert st
        for p in pieces:
            assert f_upd(st, p, len(p)) == 0
        assert f_dig(st) == want
        if ref:
            assert ref[1](pieces, s) == want
        # crc32: zlib's signature and values, running crc continues, NULL buffer gives 0
        assert lib.la_crc32(0, d, n) == zlib.crc32(d) == lib.la_crc32_host(0, d, n)
        k = rnd.randint(0, n)
        assert lib.la_crc32(lib.la_crc32(0, d[:k], k), d[k:], n - k) == zlib.crc32(d)
        if ref:
            assert ref[2](d) == zlib.crc32(d)
    assert lib.la_crc32(12345, None, 10) == 0
    # a state may also be released with free() without a digest (lz4.c:733 does that)
    st = f_init(7)
    C.CDLL(None).free(C.c_void_p(st))


def test_update_rejects_null_input():
    _, _, f_init, f_upd, f_dig = _table()
    st = f_init(0)
    assert f_upd(st, None, 0) == 1      # XXH_ERROR
    assert f_dig(st) == O.xxh32(b"", 0)


@pytest.mark.gpu
def test_large_crc32_goes_through_the_device():
    lib = _table()[0]
    rnd = random.Random(3)
    for n in (8 << 20, (8 << 20) + 12345, 300 * 1024 * 1024 + 7):
        d = rnd.randbytes(1 << 20) * (n >> 20) + rnd.randbytes(n & 0xFFFFF)
        assert len(d) == n
        seed = rnd.getrandbits(32)
        assert lib.la_crc32(seed, d, n) == zlib.crc32(d, seed) == lib.la_crc32_host(seed, d, n)
