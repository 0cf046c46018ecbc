"""GPU: the two filters behind the reference's own API shape (archive_read_open_memory2 ->
archive_read_next_header -> archive_read_data_block), checked against the oracle's
restatement of the reference filters: bytes, return code, error string, filter
code/name/count, entry metadata.  Reader block sizes 1 / 2 / 200 exercise the core's
copy-buffer paths the way the reference's tests do (test_compat_lz4.c:59)."""
import hashlib
import json
import os
import random

import pytest

import la_api
import oracle_lib as O
import streams as S

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
MANIFEST = [e for e in json.load(open(os.path.join(GOLD, "manifest.json"))) if e["codec"] in ("lz4", "gzip")]


def oracle_tuple(img, codec):
    fn = O.lz4_stream_decode if codec == "lz4" else O.gzip_stream_decode
    out, res = fn(img, 1 << 24)
    return (out.tobytes(), res.rc, res.errmsg.decode()), res


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["file"] for e in MANIFEST])
@pytest.mark.parametrize("read_size", [None, 200])
def test_reference_fixtures_through_the_api(gpu_ctx, entry, read_size):
    data = open(os.path.join(GOLD, entry["file"]), "rb").read()
    r = la_api.cat(data, read_size=read_size)
    assert r.open_rc == 0 and r.rc == la_api.ARCHIVE_EOF and r.error is None
    assert len(r.data) == entry["decoded_size"]
    assert hashlib.sha256(r.data).hexdigest() == entry["decoded_sha256"]
    code, name = (13, "lz4") if entry["codec"] == "lz4" else (1, "gzip")
    if entry["decoded_size"] or entry["codec"] == "gzip":
        assert r.filters == [(code, name), (0, "none")]          # archive_filter_code(a,0), name (test_compat_lz4.c)
        assert r.bytes_in == len(data) or entry["file"].endswith(("_2.tar.lz4", "_2.tgz"))


def test_gzip_entry_metadata_from_fixture(gpu_ctx):
    # libarchive/test/test_read_format_raw.c:122-148
    data = open(os.path.join(GOLD, "test_read_format_raw.data.gz"), "rb").read()
    for rs in (None, 1, 2):
        r = la_api.cat(data, read_size=rs)
        assert (r.data, r.pathname, r.mtime) == (b"foo\n", "test-file-name.data", 0x5CBAFD25)


@pytest.mark.parametrize("name", sorted(S.appendix_d_lz4_cases()))
def test_lz4_behaviour_table_through_the_api(gpu_ctx, name):
    img, want, rc, msg = S.appendix_d_lz4_cases()[name]
    r = la_api.cat(img)
    if r.filters and r.filters[0][1] != "lz4":
        pytest.skip("not claimed by the lz4 bidder")
    assert la_api.as_reference_tuple(r) == (want, rc, msg)


def test_lz4_small_reader_blocks_and_read_data(gpu_ctx):
    img, plain = S.synth_lz4_stream(3, 0, 5, blocks_per_frame=3, block_size=7001, nthreads=1)
    for rs in (1, 2, 200, 4096, None):
        r = la_api.cat(img.tobytes(), read_size=rs)
        assert la_api.as_reference_tuple(r) == (plain.tobytes(), 0, "")
    r = la_api.cat(img.tobytes(), use_read_data=777)
    assert r.data == plain.tobytes()


def test_lz4_multiple_batches(gpu_ctx, monkeypatch):
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    img, plain = S.synth_lz4_stream(4, 0, 40, blocks_per_frame=4, block_size=65536, nthreads=4)
    r = la_api.cat(img.tobytes(), read_size=65536)
    assert la_api.as_reference_tuple(r) == (plain.tobytes(), 0, "")
    assert len(r.block_sizes) > 3
    # an error in a late batch is reported after everything before it
    bad = bytearray(img.tobytes())
    bad[-100] ^= 0xFF
    ref, _ = oracle_tuple(bytes(bad), "lz4")
    assert la_api.as_reference_tuple(la_api.cat(bytes(bad), read_size=65536)) == ref


def test_lz4_frame_larger_than_the_window(gpu_ctx, monkeypatch):
    """ONE frame of many independent blocks, several times the gather window: the walker hands
    out the complete blocks window by window (frame records flagged OPEN / CONT) and the content
    hash state travels from window to window; output, return code and error string must be the
    reference's, also when the frame is damaged or cut somewhere in the middle."""
    import random
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(77)
    for flg in (0x74, 0x64, 0x70, 0x60):        # with / without block and content checksums
        words = [rnd.randbytes(rnd.randint(2, 9)) for _ in range(400)]
        blocks = []
        for k in range(200):
            n = rnd.choice([65536, 65536, 30000])
            d = b"".join(rnd.choice(words) for _ in range(n // 4 + 1))[:n] if k % 3 else bytes([k]) * 65536
            blocks.append((d, S.lz4_block(S.lz4_compress_block(d), bsum=bool(flg & 0x10))))
        img, plain = S.lz4_frame(blocks, flg=flg)
        assert len(img) > 3 << 20 and len(plain) > 8 << 20
        tail, tplain = S.synth_lz4_stream(9, 0, 2, blocks_per_frame=3, block_size=4096, nthreads=1)
        whole = img + tail.tobytes()
        for variant in range(5):
            m = bytearray(whole)
            if variant == 1:
                m[len(img) // 2] ^= 0x20            # damage in the middle of the big frame
            elif variant == 2:
                m = m[:len(img) * 2 // 3]           # cut inside the big frame
            elif variant == 3:
                m[len(img) - 2] ^= 0x01             # its content checksum (or last block) is wrong
            elif variant == 4:
                m = m[:len(img) - 3]                # cut inside the trailer of the big frame
            m = bytes(m)
            out, res = O.lz4_stream_decode(m, 1 << 27)
            want = (out.tobytes(), res.rc, res.errmsg.decode())
            r = la_api.cat(m, read_size=rnd.choice([None, 65536, 7777]))
            got = la_api.as_reference_tuple(r)
            assert got == want, (hex(flg), variant, len(got[0]), len(want[0]), got[1:], want[1:])
            if variant == 0:
                assert len(r.block_sizes) >= 3      # really delivered window by window


def test_lz4_window_grows_for_large_blocks(gpu_ctx, monkeypatch):
    """Frames of 1 MiB blocks behind a 1 MiB window that may grow to 4 MiB: blocks above 64 KiB
    run one per lane / wave on the device, so the filter gathers more of them per window."""
    import random
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    monkeypatch.setenv("LA_GPU_MAX_BATCH_MIB", "4")
    rnd = random.Random(5)
    frames, plain = [], b""
    for f in range(7):
        blocks = []
        for k in range(3):
            d = rnd.randbytes(rnd.choice([1 << 20, 700000, 12345]))     # incompressible: stored-size payloads
            blocks.append((d, S.lz4_block(d, stored=True, bsum=True)))
        fr, pl = S.lz4_frame(blocks, flg=0x74, bd=0x60)
        frames.append(fr)
        plain += pl
    img = b"".join(frames)
    for variant in range(3):
        m = bytearray(img)
        if variant == 1:
            m[len(m) // 2] ^= 0x08
        elif variant == 2:
            m = m[:len(m) * 3 // 5]
        m = bytes(m)
        out, res = O.lz4_stream_decode(m, 1 << 26)
        want = (out.tobytes(), res.rc, res.errmsg.decode())
        r = la_api.cat(m, read_size=65536)
        assert la_api.as_reference_tuple(r) == want, variant
        if variant == 0:
            assert out.tobytes() == plain and 2 <= len(r.block_sizes) <= 8    # a few multi-MiB windows


def test_lz4_legacy_frame_across_windows(gpu_ctx, monkeypatch):
    """A legacy frame (magic 0x184C2102, blocks of up to 8 MiB, no checksums) several windows
    long, followed by a modern frame; clean, damaged and cut."""
    import random
    import struct
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    monkeypatch.setenv("LA_GPU_MAX_BATCH_MIB", "2")
    rnd = random.Random(31)
    words = [rnd.randbytes(rnd.randint(2, 9)) for _ in range(300)]
    img, plain = S.LEGACY, b""
    for k in range(14):
        n = rnd.choice([1 << 20, 300000, 8 << 20]) if k != 13 else 12345      # the last block may be short
        d = b"".join(rnd.choice(words) for _ in range(n // 4 + 1))[:n]
        c = S.lz4_compress_block(d)
        img += struct.pack("<I", len(c)) + c
        plain += d
    tail, tplain = S.synth_lz4_stream(3, 0, 2, blocks_per_frame=2, block_size=5000, nthreads=1)
    whole = img + tail.tobytes()
    assert len(img) > 3 << 20
    for variant in range(4):
        m = bytearray(whole)
        if variant == 1:
            m[len(img) // 2] ^= 0x40
        elif variant == 2:
            m = m[:len(img) * 2 // 3]
        elif variant == 3:
            m = m[:len(img)]                     # legacy frame alone: ends at the end of the input
        m = bytes(m)
        out, res = O.lz4_stream_decode(m, 1 << 28)
        want = (out.tobytes(), res.rc, res.errmsg.decode())
        got = la_api.as_reference_tuple(la_api.cat(m, read_size=rnd.choice([None, 65536])))
        assert got == want, (variant, len(got[0]), len(want[0]), got[1:], want[1:])
        if variant == 0:
            assert got[0] == plain + tplain.tobytes()


def _dependent_frame(data, block=65536, flg=0x44):
    """A frame of DEPENDENT blocks made by liblz4's streaming compressor (every block may
    reference the 64 KiB before it)."""
    import ctypes as C
    import streams as S
    l = C.CDLL("liblz4.so.1")
    l.LZ4_createStream.restype = C.c_void_p
    l.LZ4_compress_fast_continue.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    l.LZ4_freeStream.argtypes = [C.c_void_p]
    src = C.create_string_buffer(data, len(data))
    base = C.addressof(src)
    st = l.LZ4_createStream()
    out = C.create_string_buffer(block + block // 255 + 64)
    blocks = []
    for o in range(0, len(data), block):
        n = min(block, len(data) - o)
        k = l.LZ4_compress_fast_continue(st, base + o, out, n, len(out), 1)
        assert k > 0
        blocks.append((data[o:o + n], S.lz4_block(out.raw[:k], bsum=bool(flg & 0x10))))
    l.LZ4_freeStream(st)
    return S.lz4_frame(blocks, flg=flg)


def test_lz4_dependent_frame_across_windows(gpu_ctx, monkeypatch):
    """A frame of dependent blocks several windows long: the last block of each window is the
    dictionary of the next window's first block (carried on the device), the content hash travels
    as state.  Clean, damaged, cut."""
    import random
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(91)
    words = [rnd.randbytes(rnd.randint(2, 11)) for _ in range(3000)]
    data = b"".join(rnd.choice(words) for _ in range(1600000))[:9 << 20]
    for flg in (0x44, 0x54, 0x40):
        img, plain = _dependent_frame(data, flg=flg)
        assert plain == data and len(img) > 3 << 20
        tail, tplain = S.synth_lz4_stream(2, 0, 2, blocks_per_frame=2, block_size=3000, nthreads=1)
        whole = img + tail.tobytes()
        for variant in range(4):
            m = bytearray(whole)
            if variant == 1:
                m[len(img) // 2] ^= 0x10
            elif variant == 2:
                m = m[:len(img) * 3 // 4]
            elif variant == 3:
                m[len(img) - 2] ^= 0x02
            m = bytes(m)
            out, res = O.lz4_stream_decode(m, 1 << 27)
            want = (out.tobytes(), res.rc, res.errmsg.decode())
            r = la_api.cat(m, read_size=rnd.choice([None, 65536]))
            got = la_api.as_reference_tuple(r)
            assert got == want, (hex(flg), variant, len(got[0]), len(want[0]), got[1:], want[1:])
            if variant == 0:
                assert len(r.block_sizes) >= 3


def test_lz4_file_reader(gpu_ctx, tmp_path):
    img, plain = S.synth_lz4_stream(8, 0, 6, blocks_per_frame=4, block_size=65536, nthreads=2)
    f = tmp_path / "x.lz4"
    f.write_bytes(img.tobytes())
    r = la_api.cat(None, filename=str(f))
    assert r.data == plain.tobytes() and r.filters[0] == (13, "lz4")


def _gz_cases():
    P = (b"All work and no play makes Jack a dull boy.\n" * 41)[:1800]
    import zlib
    m = S.gz_member(P)
    cases = {
        "valid": m, "level0": S.gz_member(P, level=0), "fixed": S.gz_member(P, strategy=zlib.Z_FIXED),
        "all_header_fields": S.gz_member(P, name=b"n", comment=b"c", extra=b"BC\x02\x00\x10\x00", hcrc=True),
        "bad_crc_accepted": S.gz_member(P, bad_crc=True), "bad_isize_accepted": S.gz_member(P, bad_isize=True),
        "three_members": m + S.gz_member(b"0123456789abc") + S.gz_member(b""),
        "junk_after": m + b"arbitrary junk", "junk_1f": m + b"\x1f", "junk_magic": m + b"\x1f\x8b\x08",
        "reserved_flag_second": m + b"\x1f\x8b\x08\x20" + m[4:],
        "cut_body_40": m[:40], "cut_last_body_byte": m[:-9], "trailer_4_of_8": m[:-4], "trailer_missing": m[:-8],
        "empty_member_only": S.gz_member(b""),
        "big_members": S.gz_member(os.urandom(70000)) + S.gz_member(b"z" * 200000) + S.gz_member(os.urandom(10)),
    }
    body = bytearray(S.gz_member(P * 40))
    body[300] ^= 0x10
    body[301] ^= 0x01
    cases["corrupt_body"] = bytes(body)
    cases["good_then_corrupt"] = S.gz_member(os.urandom(50000)) + S.gz_member(os.urandom(30000)) + bytes(body)
    return cases


@pytest.mark.parametrize("name", sorted(_gz_cases()))
def test_gzip_behaviour_table_through_the_api(gpu_ctx, name):
    img = _gz_cases()[name]
    ref, res = oracle_tuple(img, "gzip")
    for rs in (None, 200):
        r = la_api.cat(img, read_size=rs)
        assert la_api.as_reference_tuple(r) == ref, (name, rs)


def test_gzip_metadata_snapshot(gpu_ctx):
    def three(n):
        return b"".join(S.gz_member(bytes([65 + i]) * n, name=nm, mtime=mt)
                        for i, (nm, mt) in enumerate(((b"first", 1000), (b"second", 2000), (b"third", 3000))))
    for n, nm, mt in ((5, "third", 3000), (40000, "second", 2000)):
        r = la_api.cat(three(n))
        assert (r.pathname, r.mtime, len(r.data)) == (nm, mt, 3 * n)
    img = S.gz_member(bytes(65536), name=b"first", mtime=1000) + S.gz_member(b"x", name=b"second", mtime=2000)
    r = la_api.cat(img)
    assert (r.pathname, r.mtime, len(r.data)) == ("first", 1000, 65537)


def test_gzip_strict_mode_rejects_bad_crc(gpu_ctx, monkeypatch):
    monkeypatch.setenv("LA_GZIP_STRICT", "1")
    P = b"payload " * 500
    r = la_api.cat(S.gz_member(b"ok" * 10) + S.gz_member(P, bad_crc=True))
    assert r.rc == la_api.ARCHIVE_FATAL and "CRC32" in r.error


def test_gzip_mutated_streams(gpu_ctx):
    rnd = random.Random(6)
    base = S.gz_member(os.urandom(3000) + b"abc" * 3000, name=b"a") + S.gz_member(b"hello world " * 700) + S.gz_member(b"tail")
    for t in range(40):
        m = bytearray(base)
        for _ in range(rnd.randint(1, 3)):
            m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
        if rnd.random() < 0.3:
            m = m[:rnd.randrange(1, len(m))]
        m = bytes(m)
        ref, _ = oracle_tuple(m, "gzip")
        r = la_api.cat(m)
        if r.filters and r.filters[0][1] != "gzip":
            continue
        assert la_api.as_reference_tuple(r) == ref, t


def test_gzip_members_with_unusual_xfl_os_bytes(gpu_ctx, monkeypatch):
    """The boundary search first trusts only headers with the XFL / OS bytes real writers emit
    (LA_GZ_INDEX_STRICT).  A stream whose headers carry other values (the reference accepts any,
    gzip.c:128-239) must still come out whole: the filter notices the passed-over header behind the
    first member's trailer and goes on with every 1f 8b 08 as a candidate."""
    import random
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(313)
    words = [rnd.randbytes(rnd.randint(2, 9)) for _ in range(400)]
    for xfl, osb, every in ((7, 77, 1), (0, 200, 1), (9, 3, 3)):
        parts, plain = [], b""
        for i in range(70):
            d = b"".join(rnd.choice(words) for _ in range(rnd.randint(1, 12000)))
            m = bytearray(S.gz_member(d, level=rnd.choice([1, 6, 9])))
            if i % every == 0:
                m[8], m[9] = xfl, osb
            parts.append(bytes(m))
            plain += d
        img = b"".join(parts)
        for tail in (b"", b"trailing junk that is not a member"):
            out, res = O.gzip_stream_decode(img + tail, len(plain) + 64)
            assert out.tobytes() == plain and res.rc == 0
            r = la_api.cat(img + tail, read_size=rnd.choice([None, 4096]))
            assert la_api.as_reference_tuple(r) == (plain, 0, ""), (xfl, osb, every, len(r.data), r.error)


def _bgzf_member(data, bsize_delta=0, level=6):
    """A gzip member with the BGZF-compatible 'BC' size subfield (total member size - 1, u16): the C3
    stream shape.  bsize_delta != 0 writes a WRONG size into the subfield."""
    import struct
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9)
    body = co.compress(data) + co.flush()
    total = 18 + len(body) + 8
    hdr = (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\x03" + struct.pack("<H", 6) + b"BC" +
           struct.pack("<HH", 2, (total - 1 + bsize_delta) & 0xFFFF))
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def test_gzip_bgzf_indexed_members_through_the_filter(gpu_ctx, monkeypatch):
    """C3's member shape driven through the gzip FILTER (bid, header parse, indexed boundaries, decode,
    CRC32 / ISIZE) on the device, against the oracle: correct BSIZE subfields, then members whose
    subfield lies (too small, too large, pointing past the end) -- the subfield is only a hint, the
    decode decides where a member ends, exactly as the reference (which ignores FEXTRA, gzip.c:185-199)."""
    rnd = random.Random(99)
    words = [rnd.randbytes(rnd.randint(2, 11)) for _ in range(300)]
    datas = [b"".join(rnd.choice(words) for _ in range(rnd.randint(1, 9000)))[:65280] for _ in range(48)]
    plain = b"".join(datas)
    good = b"".join(_bgzf_member(d, level=rnd.choice([1, 6, 9])) for d in datas)
    for batch in ("1", None):
        if batch:
            monkeypatch.setenv("LA_GPU_BATCH_MIB", batch)
        else:
            monkeypatch.delenv("LA_GPU_BATCH_MIB", raising=False)
        ref, _ = oracle_tuple(good, "gzip")
        assert ref == (plain, 0, "")
        for rs in (None, 4096):
            assert la_api.as_reference_tuple(la_api.cat(good, read_size=rs)) == ref
        # lying subfields on some members
        for deltas in ((0, -5, 0, 0, 7), (3,), (0, 0, 0, 0, 0, 0, 40000), (-17, 0, 25)):
            img = b"".join(_bgzf_member(d, bsize_delta=deltas[i % len(deltas)]) for i, d in enumerate(datas))
            ref, _ = oracle_tuple(img, "gzip")
            assert ref == (plain, 0, "")
            assert la_api.as_reference_tuple(la_api.cat(img)) == ref, deltas
        # damaged / cut streams of indexed members: bytes before the event, rc and message as the oracle
        for cut in (len(good) - 3, len(good) // 2, 19):
            ref, _ = oracle_tuple(good[:cut], "gzip")
            assert la_api.as_reference_tuple(la_api.cat(good[:cut])) == ref, cut
        bad = bytearray(good)
        bad[len(good) // 3] ^= 0x5A
        ref, _ = oracle_tuple(bytes(bad), "gzip")
        assert la_api.as_reference_tuple(la_api.cat(bytes(bad))) == ref


def test_gzip_single_member_across_windows(gpu_ctx, monkeypatch):
    """ONE member several times the gather window (BASELINE configs[0] shape, small): the window has to
    grow until the member fits; output, rc and the entry name as the oracle says."""
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(12)
    words = [rnd.randbytes(rnd.randint(3, 14)) for _ in range(500)]
    plain = b"".join(rnd.choice(words) for _ in range(700000))[:5 * 1024 * 1024 + 123]
    img = S.gz_member(plain, name=b"one.bin", mtime=77)
    assert len(img) > 1024 * 1024        # larger than the 1 MiB gather window
    ref, _ = oracle_tuple(img, "gzip")
    assert ref == (plain, 0, "")
    r = la_api.cat(img, read_size=65536)
    assert la_api.as_reference_tuple(r) == ref and r.pathname == "one.bin" and r.mtime == 77
    ref, _ = oracle_tuple(img[:-100000], "gzip")
    assert la_api.as_reference_tuple(la_api.cat(img[:-100000])) == ref


def test_lz4_window_bounded_by_decoded_bytes(gpu_ctx, monkeypatch):
    """Highly compressible input: a window of compressed bytes would decode to far more than a sane slab
    (a few MiB of zeros ask for GiBs).  LA_GPU_OUT_BUDGET_MIB bounds a window by decoded bytes too: the
    walker stops in front of the block that would pass it -- also INSIDE a frame (independent blocks with
    a content checksum carried over, dependent blocks with their history, legacy blocks) -- and the next
    window goes on there.  The byte stream, rc and message stay the reference's."""
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_OUT_BUDGET_MIB", "1")        # 16 blocks of 64 KiB per window
    rnd = random.Random(5150)
    zeros = [(bytes(65536), S.lz4_block(S.lz4_compress_block(bytes(65536)), bsum=True)) for _ in range(70)]
    f1, p1 = S.lz4_frame(zeros, flg=0x74)                   # independent, block + content checksums
    words = [rnd.randbytes(rnd.randint(2, 9)) for _ in range(50)]
    chunks = [b"".join(rnd.choice(words) for _ in range(9000))[:rnd.choice([65536, 65536, 1234])] for _ in range(45)]
    f2, p2 = S.lz4_dependent_frame(chunks)                  # dependent blocks across windows
    tail, tplain = S.synth_lz4_stream(3, 0, 40, blocks_per_frame=2, block_size=65536, nthreads=2)
    img = f1 + f2 + tail.tobytes()
    plain = p1 + p2 + tplain.tobytes()
    assert len(img) < 6 << 20 and len(plain) > 10 << 20
    for variant in range(3):
        m = bytearray(img)
        if variant == 1:
            m[len(f1) - 2] ^= 0x40          # the first frame's content checksum is wrong
        elif variant == 2:
            m = m[:len(f1) + len(f2) // 2]  # cut inside the dependent frame
        m = bytes(m)
        out, res = O.lz4_stream_decode(m, 1 << 27)
        want = (out.tobytes(), res.rc, res.errmsg.decode())
        if variant == 0:
            assert want == (plain, 0, "")
        r = la_api.cat(m, read_size=rnd.choice([None, 65536]))
        assert la_api.as_reference_tuple(r) == want, variant
        if variant == 0:
            assert len(r.block_sizes) >= 8  # many bounded windows, not one huge slab


def test_gzip_bid_only_indexed_switch(gpu_ctx, monkeypatch):
    """LA_GZIP_BID_ONLY_INDEXED=1: the bidder declines streams without the BGZF size subfield (a deployment
    leaves those to the reference's bidder); with no other gzip bidder registered here the bytes then come
    through unfiltered.  BGZF-style streams are taken as before."""
    plain = b"some text " * 1000
    ordinary, indexed = S.gz_member(plain), _bgzf_member(plain)
    r = la_api.cat(ordinary)
    assert r.filters[0] == (1, "gzip") and r.data == plain
    monkeypatch.setenv("LA_GZIP_BID_ONLY_INDEXED", "1")
    r = la_api.cat(ordinary)
    assert r.filters[0][1] != "gzip" and r.data == ordinary
    r = la_api.cat(indexed)
    assert r.filters[0] == (1, "gzip") and r.data == plain


def test_gzip_window_bounded_by_decoded_bytes(gpu_ctx, monkeypatch):
    """Highly compressible members: one window of compressed bytes would claim far more slab than is sane (deflate
    expands up to 1032 x).  LA_GPU_OUT_BUDGET_MIB bounds a window by the decoded bytes its members ask for; the rest of
    the window is decoded by the following windows.  Bytes, rc and message stay the reference's."""
    monkeypatch.setenv("LA_GPU_OUT_BUDGET_MIB", "1")
    rnd = random.Random(808)
    members = [S.gz_member(bytes(rnd.choice([65536, 200000, 10]))) for _ in range(60)] + [S.gz_member(b"tail " * 1000)]
    img = b"".join(members)
    plain_len = sum(len(__import__("zlib").decompress(m, 31)) for m in members)
    assert len(img) < 100000 and plain_len > 3 << 20
    for variant in range(3):
        m = img if variant == 0 else img[:len(img) // 2] if variant == 1 else img[:700] + b"\x00" + img[701:]
        ref, _ = oracle_tuple(m, "gzip")
        r = la_api.cat(m, read_size=rnd.choice([None, 4096]))
        assert la_api.as_reference_tuple(r) == ref, variant
        if variant == 0:
            assert len(ref[0]) == plain_len and len(r.block_sizes) >= 4     # several bounded windows


def test_lz4_stream_across_the_window_ramp(gpu_ctx, monkeypatch):
    """Default windows: the first one holds 16 MiB of the stream, the next 32, then 64 (la_filter_lz4.c, init).  A
    C2-shaped stream of 150 MiB decoded (about 66 MiB compressed) passes all three sizes; frames straddle every window
    edge (a frame is 1 MiB decoded, about 450 KiB compressed: no window size is a multiple of it)."""
    monkeypatch.delenv("LA_GPU_BATCH_MIB", raising=False)
    img, plain = S.synth_lz4_stream(0x52414D50, 0, 150, nthreads=8)
    assert img.size > (48 << 20)
    out = la_api.cat(img.tobytes(), read_size=1 << 20)
    got, rc, msg = la_api.as_reference_tuple(out)
    assert (rc, msg) == (0, "") and len(got) == plain.size
    assert got == plain.tobytes()


def test_gzip_indexed_members_copy_ahead_across_many_windows(gpu_ctx, monkeypatch):
    """The gzip filter's second slab (round 3): for a window of indexed members the decoded bytes are copied towards the
    OTHER slab behind the decode, at the offset the carry will take, and the slabs change roles when the window came out
    as indexed.  Members of every size (so that the held-back partial 64 KiB block -- the carry -- has every length),
    several 1 MiB windows, small and large reads; then the same stream with a lying size subfield in the middle (the
    copy-ahead is dropped for the retry) and with a damaged member (bytes before it, then the reference's error)."""
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(4242)
    words = [rnd.randbytes(rnd.randint(2, 12)) for _ in range(500)]
    datas = []
    for i in range(420):
        n = rnd.choice([1, 7, 100, 4000, 20000, 65280, 65280, 65280, rnd.randint(1, 65280)])
        datas.append(b"".join(rnd.choice(words) for _ in range(n // 4 + 1))[:n])
    plain = b"".join(datas)
    good = b"".join(_bgzf_member(d, level=rnd.choice([1, 6])) for d in datas)
    assert len(good) > (4 << 20)
    ref, _ = oracle_tuple(good, "gzip")
    assert ref == (plain, 0, "")
    for rs in (None, 4096, 1000003):
        assert la_api.as_reference_tuple(la_api.cat(good, read_size=rs)) == ref, rs
    lying = b"".join(_bgzf_member(d, bsize_delta=(-9 if i == 200 else 31 if i == 333 else 0)) for i, d in enumerate(datas))
    ref, _ = oracle_tuple(lying, "gzip")
    assert ref == (plain, 0, "")
    assert la_api.as_reference_tuple(la_api.cat(lying, read_size=65536)) == ref
    bad = bytearray(good)
    bad[len(good) * 2 // 3] ^= 0x81
    ref, _ = oracle_tuple(bytes(bad), "gzip")
    assert la_api.as_reference_tuple(la_api.cat(bytes(bad))) == ref
