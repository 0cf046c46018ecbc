"""zstd read filter through the reference's API shape: archive_read_support_filter_all ->
bidder -> la_filter_zstd.c -> la_gpu_zstd_decode (C ABI) -> archive_read_data_block, against the reference's own zstd
fixtures, the oracle (oracle/orc_zstd.c) and the image's libzstd (the library the reference's filter calls)."""
import hashlib
import json
import os
import random

import pytest

import la_api
import zstd_support as Z

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
MANIFEST = [e for e in json.load(open(os.path.join(GOLD, "manifest.json"))) if e["codec"] == "zstd"]
ARCHIVE_FILTER_ZSTD = 14


def _z():
    z = Z.libzstd()
    if z is None:
        pytest.skip("no libzstd.so.1 in this image (test inputs are made with it)")
    return z


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["file"] for e in MANIFEST])
def test_zstd_reference_fixtures_through_the_api(gpu_ctx, entry):
    data = open(os.path.join(GOLD, entry["file"]), "rb").read()
    res = la_api.cat(data)
    assert res.rc in (la_api.ARCHIVE_EOF, la_api.ARCHIVE_OK), res.error
    if entry["decoded_size"]:
        assert res.filters[0] == (ARCHIVE_FILTER_ZSTD, "zstd")     # test_compat_zstd.c:67-68
    assert len(res.data) == entry["decoded_size"]
    assert hashlib.sha256(res.data).hexdigest() == entry["decoded_sha256"]


def test_zstd_compat_tars_list_like_the_reference_test(gpu_ctx):
    # test_compat_zstd.c:37-70: six entries, ustar, filter code / name
    for name in ("test_compat_zstd_1.tar.zst", "test_compat_zstd_2.tar.zst"):
        res = la_api.list_entries(open(os.path.join(GOLD, name), "rb").read())
        assert [e[0] for e in res.entries] == ["f1", "f2", "f3", "d1/f1", "d1/f2", "d1/f3"]
        assert res.rc == la_api.ARCHIVE_EOF
        assert res.filters[0] == (ARCHIVE_FILTER_ZSTD, "zstd")


@pytest.mark.parametrize("lane_kernel", [0, 1], ids=["wave-per-frame", "lane-per-frame"])
def test_zstd_every_level_and_shape(gpu_ctx, monkeypatch, lane_kernel):
    """both device kernels (LA_ZSTD_OPT_LANE_KERNEL) against the plain bytes and the oracle"""
    monkeypatch.setenv("LA_ZSTD_LANE_KERNEL", str(lane_kernel))
    z, o = _z(), Z.oracle_lib()
    rnd = random.Random(0x2A)
    for it in range(60):
        n = rnd.choice([1, 5, 100, 1000, 5000, 70000, 200000, 400000])
        d = Z.gen(rnd, n, rnd.randint(0, 4))
        img = Z.zstd_compress(z, d, rnd.choice([-5, 1, 3, 5, 9, 15, 19]))
        res = la_api.cat(img)
        assert la_api.as_reference_tuple(res) == (d, 0, ""), (it, n, res.error)
        assert Z.oracle_decode(o, img, n + 16) == (0, d, "")


def test_zstd_many_frames_with_skippable_frames_and_small_reads(gpu_ctx, monkeypatch):
    """pzstd shape: thousands of frames; skippable frames between them; upstream hands out 1000-byte pieces; windows of
    1 MiB of compressed input."""
    z = _z()
    rnd = random.Random(9)
    parts, plain = [], []
    for i in range(1500):
        d = Z.gen(rnd, rnd.randint(0, 12000), rnd.randint(1, 4))
        parts.append(Z.zstd_compress(z, d, rnd.choice([1, 3, 9])))
        plain.append(d)
        if i % 97 == 5:
            parts.append(Z.skippable(b"s" * (i % 50), i % 16))
    img, want = b"".join(parts), b"".join(plain)
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    res = la_api.cat(img, read_size=1000)
    assert la_api.as_reference_tuple(res) == (want, 0, "")
    assert res.bytes_in == len(img) and res.bytes_out == len(want)


def test_zstd_stream_that_starts_with_a_skippable_frame(gpu_ctx):
    # the zstd bidder takes 0x184D2A5x (zstd.c:117-130); libzstd skips the frame
    z = _z()
    d = b"after the skippable frame\n" * 10
    res = la_api.cat(Z.skippable(b"meta") + Z.zstd_compress(z, d, 3))
    assert la_api.as_reference_tuple(res) == (d, 0, "")
    assert res.filters[0] == (ARCHIVE_FILTER_ZSTD, "zstd")


@pytest.mark.parametrize("lane_kernel", [0, 1], ids=["wave-per-frame", "lane-per-frame"])
def test_zstd_truncated_and_damaged_streams(gpu_ctx, monkeypatch, lane_kernel):
    """Frames in front of the damage are delivered, then ARCHIVE_FATAL with the reference's strings (zstd.c:213-217,
    :226-231).  The verdict (accept / refuse) is the oracle's; where the stream is accepted the bytes are libzstd's."""
    monkeypatch.setenv("LA_ZSTD_LANE_KERNEL", str(lane_kernel))
    z, o = _z(), Z.oracle_lib()
    rnd = random.Random(0x99)
    refused = 0
    for it in range(120):
        frames = [Z.gen(rnd, rnd.choice([50, 3000, 70000]), rnd.randint(1, 4)) for _ in range(3)]
        imgs = [Z.zstd_compress(z, d, rnd.choice([1, 3, 19])) for d in frames]
        img = bytearray(b"".join(imgs))
        good_prefix = frames[0] + frames[1]
        lo = len(imgs[0]) + len(imgs[1])
        trunc = it % 3 == 0
        if trunc:
            img = img[:rnd.randint(lo + 1, len(img) - 1)]
        else:
            img[rnd.randrange(lo, len(img))] ^= 1 << rnd.randrange(8)
        img = bytes(img)
        rc, out, msg = Z.oracle_decode(o, img, 400000)
        res = la_api.cat(img)
        data, grc, gmsg = la_api.as_reference_tuple(res)
        if rc == 0:
            assert (data, grc) == (out, 0), it
            continue
        refused += 1
        assert grc == la_api.ARCHIVE_FATAL, it
        assert data == good_prefix, it
        if trunc:
            assert msg == "Truncated zstd input"
        if msg == "Truncated zstd input":     # (also a damaged size field that points past the end of the input)
            assert gmsg == msg, it
        else:
            assert gmsg.startswith("Zstd decompression failed: "), gmsg
    assert refused > 60


def test_zstd_garbage_behind_a_frame(gpu_ctx):
    # libzstd: "Unknown frame descriptor" for bytes that are no frame where one must start
    z = _z()
    d = b"x" * 1000
    res = la_api.cat(Z.zstd_compress(z, d, 3) + b"garbage!")
    assert la_api.as_reference_tuple(res) == (d, la_api.ARCHIVE_FATAL, "Zstd decompression failed: Unknown frame descriptor")


def test_zstd_frame_whose_blocks_claim_more_than_the_window_budget(gpu_ctx):
    """40 000 RLE blocks of 128 KiB (5 GiB of one byte in 160 KB of input): the reference streams it; this data plane
    decodes whole frames into HBM and must refuse it by name instead of asking for the memory (found by the ASan fuzz)."""
    hdr = (0xFD2FB528).to_bytes(4, "little") + bytes([0x00, 0x70])       # no content size, window descriptor
    blocks = bytearray()
    nb = 40000
    for i in range(nb):
        bh = (1 if i == nb - 1 else 0) | (1 << 1) | ((128 * 1024) << 3)      # last?, RLE, Block_Size
        blocks += bh.to_bytes(3, "little") + b"z"
    res = la_api.cat(hdr + bytes(blocks))
    assert res.rc == la_api.ARCHIVE_FATAL
    assert res.error.startswith("zstd frame too large for the GPU data plane"), res.error


def test_zstd_one_frame_larger_than_the_gather_limit(gpu_ctx, monkeypatch):
    """a frame of 3 MiB of raw blocks with the gather limit at 1 MiB: refused by name, not gathered without bound;
    the same frame passes with the default limit"""
    hdr = (0xFD2FB528).to_bytes(4, "little") + bytes([0x00, 0x70])
    body = bytearray()
    nb = 24
    for i in range(nb):
        body += ((1 if i == nb - 1 else 0) | (0 << 1) | ((128 * 1024) << 3)).to_bytes(3, "little") + bytes([65 + i]) * (128 * 1024)
    img = hdr + bytes(body)
    want = b"".join(bytes([65 + i]) * (128 * 1024) for i in range(nb))
    assert la_api.as_reference_tuple(la_api.cat(img, read_size=65536)) == (want, 0, "")
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    monkeypatch.setenv("LA_GPU_MAX_BATCH_MIB", "1")
    res = la_api.cat(img, read_size=65536)
    assert res.rc == la_api.ARCHIVE_FATAL and res.error.startswith("zstd frame too large for the GPU data plane (more than"), res.error


def test_zstd_frame_of_compressed_blocks_beyond_the_window_budget(gpu_ctx):
    """The same refusal for a frame made of COMPRESSED blocks (not RLE): 40 000 blocks of {one raw literal, no
    sequences} -- three bytes each, one decoded byte each, but a compressed block may decode to 128 KiB and the header
    carries no content size, so the walker must reserve 5 GiB for the frame: refused by name.  A hundred of the same
    blocks decode bit for bit (the block really is a valid compressed block)."""
    hdr = (0xFD2FB528).to_bytes(4, "little") + bytes([0x00, 0x70])       # no content size, window descriptor

    def frame(nb):
        body = bytearray()
        for i in range(nb):
            bh = (1 if i == nb - 1 else 0) | (2 << 1) | (3 << 3)             # last?, Compressed_Block, Block_Size 3
            body += bh.to_bytes(3, "little") + bytes([0x08, 0x78, 0x00])     # raw literals of size 1 ("x"), 0 sequences
        return hdr + bytes(body)

    assert la_api.as_reference_tuple(la_api.cat(frame(100))) == (b"x" * 100, 0, "")
    res = la_api.cat(frame(40000))
    assert res.rc == la_api.ARCHIVE_FATAL
    assert res.error.startswith("zstd frame too large for the GPU data plane (its blocks may decode to"), res.error
