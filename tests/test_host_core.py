"""CPU tests of the host side: the C ABI surface (every symbol the headers declare is
exported), the minimal read core's peek/consume contracts with small reader blocks
(the reference's own trick, test_read_format_raw.c:77 / test_compat_lz4.c:59), the
walkers, and the loud failure of the filters when no GPU is present."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import la_api
import streams as S
import libarchive_amd as la
from libarchive_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(la_[a-z0-9_]+|archive_[a-z0-9_]+)\s*\(", text))
    names -= {"la_rc"}
    # drop typedef'd callback types
    return {n for n in names if not n.endswith("_callback")}


@pytest.mark.parametrize("header,lib", [("la_gpu.h", "gpu"), ("la_host.h", "host"), ("la_archive.h", "host")])
def test_every_declared_symbol_is_exported(header, lib):
    handle = la.gpu_lib() if lib == "gpu" else la.host_lib()
    missing = [n for n in sorted(_declared_functions(header)) if not hasattr(handle, n)]
    assert missing == []


def test_abi_struct_sizes():
    assert N.LZ4_BLOCK_DTYPE.itemsize == 24 and N.LZ4_FRAME_DTYPE.itemsize == 32
    assert N.GZ_MEMBER_DTYPE.itemsize == 24 and N.GZ_RESULT_DTYPE.itemsize == 16
    assert la.gpu_lib().la_gpu_abi_version() == 3


@pytest.mark.parametrize("read_size", [1, 2, 7, 200, 65536, None])
def test_raw_passthrough_with_small_reader_blocks(read_size):
    rnd = random.Random(read_size or 0)
    data = rnd.randbytes(5000)
    r = la_api.cat(data, read_size=read_size)
    assert r.open_rc == 0 and r.rc == la_api.ARCHIVE_EOF
    assert r.data == data and r.pathname == "data" and r.format_name == "raw"
    assert r.filters == [(0, "none")]
    assert r.bytes_in == len(data) == r.bytes_out
    r2 = la_api.cat(data, read_size=read_size, use_read_data=333)
    assert r2.data == data


def test_empty_input_selects_empty_format():
    r = la_api.cat(b"")
    assert r.open_rc == 0 and r.rc == la_api.ARCHIVE_EOF and r.data == b""


def test_file_client(tmp_path):
    data = os.urandom(300000)
    f = tmp_path / "plain.bin"
    f.write_bytes(data)
    r = la_api.cat(None, filename=str(f))
    assert r.data == data and max(r.block_sizes) <= 65536     # block size rounded up to 64 KiB (open_filename.c:388-396)


def test_filters_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    img, _ = S.lz4_frame([(S.P38, S.lz4_block(S.P38, stored=True))])
    r = la_api.cat(img)
    assert r.open_rc == la_api.ARCHIVE_FATAL and "GPU data plane" in r.error and "no CPU fallback" in r.error
    r = la_api.cat(S.gz_member(b"hello"))
    assert r.open_rc == la_api.ARCHIVE_FATAL and "GPU data plane" in r.error
    with pytest.raises(RuntimeError):
        la.GpuContext(0)


def test_bidders_ignore_what_the_reference_ignores():
    # first header with reserved bits / wrong version: no bid, the file passes through raw (F11 ii)
    for img in (S.MAGIC + bytes([0x66, 0x40]) + bytes(20), S.MAGIC + bytes([0x64, 0x30]) + bytes(20),
                b"\x1f\x8b\x08\x20" + bytes(30), b"\x1f\x8b\x07\x00" + bytes(30), S.MAGIC[:3] + b"\x00" * 20):
        r = la_api.cat(img)
        assert r.open_rc == 0 and r.filters == [(0, "none")] and r.data == img
    short = (S.MAGIC + S.lz4_desc(0x64, 0x40))[:5]
    r = la_api.cat(short)
    assert r.data == short                                     # bidder needs 11 bytes (lz4.c:150)


def test_gzip_walker_bgzf_and_speculative():
    host = la.host_lib()
    host.la_gz_index_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    host.la_gz_index_free.argtypes = [C.c_void_p]

    class Idx(C.Structure):
        _fields_ = [("members", C.c_void_p), ("headers", C.c_void_p), ("n", C.c_uint32), ("cap", C.c_uint32),
                    ("end_kind", C.c_int), ("consumed", C.c_uint64), ("max_out", C.c_uint64), ("speculative", C.c_int)]

    def walk(img, at_eof=1):
        buf = np.frombuffer(img, dtype=np.uint8).copy()
        x = Idx()
        assert host.la_gz_index_build(buf.ctypes.data, buf.size, at_eof, C.byref(x)) == 0
        mem = np.empty(x.n, dtype=N.GZ_MEMBER_DTYPE)
        if x.n:
            C.memmove(mem.ctypes.data, x.members, mem.nbytes)
        out = (x.n, x.end_kind, x.consumed, x.max_out, x.speculative, mem)
        host.la_gz_index_free(C.byref(x))
        return out

    parts = [os.urandom(1000) * 3, b"abc" * 5000, b""]
    plain = [S.gz_member(p, name=b"n%d" % i) for i, p in enumerate(parts)]
    img = b"".join(plain)
    n, end, consumed, max_out, spec, mem = walk(img)
    assert n == 3 and end == N.LA_END_EOF and spec == 1 and consumed == len(img)
    assert list(mem["dst_cap"]) == [len(p) for p in parts] and max_out == sum(len(p) for p in parts)
    # trailing bytes stay inside the last (speculative) span: the decode finds the real end
    n, end, consumed, _, spec, mem = walk(img + b"trailing junk")
    assert n == 3 and end == N.LA_END_EOF and consumed == len(img) + 13
    assert list(mem["dst_cap"][:2]) == [len(p) for p in parts[:2]]
    # BGZF-style members carry their own size: exact boundaries
    def bgzf(p):
        m = S.gz_member(p, extra=b"BC\x02\x00\x00\x00")
        total = len(m)
        return m[:16] + (total - 1).to_bytes(2, "little") + m[18:]
    img2 = b"".join(bgzf(p) for p in parts)
    n, end, consumed, max_out, spec, mem = walk(img2)
    assert n == 3 and spec == 0 and consumed == len(img2) and list(mem["dst_cap"]) == [len(p) for p in parts]
    # a window that ends inside the last member asks for more input
    n, end, consumed, _, _, _ = walk(img2[:-10], at_eof=0)
    assert end == N.LA_END_NEED_MORE and n == 2


_SCAN_DRIVER = r'''
import random, sys, os
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libarchive_amd as la
import streams as S
rnd = random.Random(11)
total = 0
for t in range(120):
    parts = []
    for k in range(rnd.randint(1, 12)):
        d = rnd.randbytes(rnd.choice([0, 1, 30, 31, 32, 33, 63, 64, 65, 500, 5000]))
        m = bytearray(S.gz_member(d, level=rnd.choice([0, 6])))
        if rnd.random() < 0.4 and len(m) > 40:        # false 1f 8b 08 (and near misses) inside the deflate data
            q = rnd.randrange(12, len(m) - 12)
            m[q:q + 3] = rnd.choice([b"\x1f\x8b\x08", b"\x1f\x8b\x07", b"\x1f\x1f\x8b", b"\x1f\x8b\x08"])
            m[q + 3] = rnd.choice([0, 0, 8, 0x20])
        parts.append(bytes(m))
    img = np.frombuffer(b"".join(parts) + rnd.randbytes(rnd.randint(0, 40)), dtype=np.uint8)
    for at_eof in (True, False):
        idx = la.gz_index(img, at_eof=at_eof)
        print(t, at_eof, idx.end_kind, idx.consumed, [(int(m["src_off"]), int(m["src_len"])) for m in idx.members])
        total += len(idx.members)
print("members", total)
'''


def test_gzip_boundary_scan_vector_and_fallback_agree(tmp_path):
    """The member-boundary search has an AVX2 two-byte compare and a memchr fallback (LA_NO_AVX2=1): the same
    member tables from both, on streams with false magics and near misses planted at random alignments."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "scan.py"
    script.write_text("ROOT = %r\n" % root + _SCAN_DRIVER)
    outs = []
    for no_avx2 in (False, True):
        env = dict(os.environ)
        if no_avx2:
            env["LA_NO_AVX2"] = "1"
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-3000:]
        outs.append(out.stdout)
    assert outs[0] == outs[1]
    assert int(outs[0].strip().rsplit(" ", 1)[1]) > 500
