"""The bid policy of the three GPU read filters (host/la_bid_policy.c), through the reference's API shape
(la_api.cat = bsdcat): a stream of MANY independent units is taken, ONE large serial unit -- a plain single-member
.gz (gzip.c:431-511), a single lz4 frame with a content checksum (lz4.c:615-668), a one-frame .zst
(zstd.c:196-260) -- is NOT bid for, so that libarchive's own filter, registered beside this one, decodes it
(archive_read.c:557-565 takes the highest bid; with no other bidder the bytes pass through raw, which is what these
tests observe).  The same functions run on the CPU against tests/mock_gpu (test_host_filters_mock.py)."""
import random
import struct
import zlib

import pytest

import la_api
import streams as S

pytestmark = pytest.mark.gpu

ARCHIVE_FILTER_NONE, ARCHIVE_FILTER_GZIP, ARCHIVE_FILTER_LZ4, ARCHIVE_FILTER_ZSTD = 0, 1, 13, 14


def _codes(res):
    return [c for c, _ in res.filters]


def _gz(data, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    return co.compress(data) + co.flush()


def test_gzip_large_single_member_is_declined_many_members_are_taken(gpu_ctx, monkeypatch):
    rnd = random.Random(5)
    big = rnd.randbytes(700_000)                       # one member, > 256 KiB compressed
    lone = _gz(big, 1)
    many_plain = b"".join(rnd.randbytes(20_000) for _ in range(40))
    many = b"".join(_gz(many_plain[i:i + 20_000], 1) for i in range(0, len(many_plain), 20_000))
    small_plain = b"The quick brown fox jumps over the lazy dog. " * 200
    small = _gz(small_plain)
    monkeypatch.setenv("LA_GPU_BID", "auto")
    r = la_api.cat(lone)
    assert ARCHIVE_FILTER_GZIP not in _codes(r) and r.data == lone, "a lone 700 KB member must be left to the CPU filter"
    r = la_api.cat(many)
    assert ARCHIVE_FILTER_GZIP in _codes(r) and r.data == many_plain
    r = la_api.cat(small)
    assert ARCHIVE_FILTER_GZIP in _codes(r) and r.data == small_plain
    # the look-ahead is a knob: with 1 MiB of it the 700 KB stream is "small" and taken
    monkeypatch.setenv("LA_GPU_BID_LOOKAHEAD_KIB", "1024")
    r = la_api.cat(lone)
    assert ARCHIVE_FILTER_GZIP in _codes(r) and r.data == big
    monkeypatch.delenv("LA_GPU_BID_LOOKAHEAD_KIB")
    monkeypatch.setenv("LA_GPU_BID", "all")
    r = la_api.cat(lone)
    assert ARCHIVE_FILTER_GZIP in _codes(r) and r.data == big


def test_lz4_single_frame_with_content_checksum_is_declined(gpu_ctx, monkeypatch):
    rnd = random.Random(6)
    plain = rnd.randbytes(24 * 65536)                  # 1.5 MiB of stored blocks: the frame ends behind the look-ahead
    blocks = [(plain[i:i + 65536], S.lz4_block(plain[i:i + 65536], stored=True)) for i in range(0, len(plain), 65536)]
    with_sum, _ = S.lz4_frame(blocks, flg=0x64)        # independent blocks + content checksum
    without, _ = S.lz4_frame(blocks, flg=0x60)         # no content checksum: block-parallel whatever its size
    frames = b"".join(S.lz4_frame(blocks[i:i + 4], flg=0x64)[0] for i in range(0, len(blocks), 4))   # six 256 KiB frames
    monkeypatch.setenv("LA_GPU_BID", "auto")
    r = la_api.cat(with_sum)
    assert ARCHIVE_FILTER_LZ4 not in _codes(r) and r.data == with_sum
    r = la_api.cat(without)
    assert ARCHIVE_FILTER_LZ4 in _codes(r) and r.data == plain
    r = la_api.cat(frames)
    assert ARCHIVE_FILTER_LZ4 in _codes(r) and r.data == plain
    monkeypatch.setenv("LA_GPU_BID", "all")
    r = la_api.cat(with_sum)
    assert ARCHIVE_FILTER_LZ4 in _codes(r) and r.data == plain


def _zstd_raw_frame(data):
    """one zstd frame of raw blocks (no compression needed for the walker): magic, single-segment header with a
    4-byte content size, raw blocks of at most 128 KiB"""
    out = bytearray(struct.pack("<IB", 0xFD2FB528, 0xA0) + struct.pack("<I", len(data)))   # FCS flag 2 + single segment
    n = len(data)
    pos = 0
    while True:
        k = min(131072, n - pos)
        last = pos + k >= n
        hdr = (k << 3) | (0 << 1) | (1 if last else 0)
        out += struct.pack("<I", hdr)[:3] + data[pos:pos + k]
        pos += k
        if last:
            break
    return bytes(out)


def test_zstd_one_large_frame_is_declined_small_frames_are_taken(gpu_ctx, monkeypatch):
    rnd = random.Random(7)
    plain = rnd.randbytes(1_500_000)
    one = _zstd_raw_frame(plain)
    many = b"".join(_zstd_raw_frame(plain[i:i + 100_000]) for i in range(0, len(plain), 100_000))
    monkeypatch.setenv("LA_GPU_BID", "auto")
    r = la_api.cat(one)
    assert ARCHIVE_FILTER_ZSTD not in _codes(r) and r.data == one
    r = la_api.cat(many)
    assert ARCHIVE_FILTER_ZSTD in _codes(r) and r.data == plain
    monkeypatch.setenv("LA_GPU_BID", "all")
    r = la_api.cat(one)
    assert ARCHIVE_FILTER_ZSTD in _codes(r) and r.data == plain
