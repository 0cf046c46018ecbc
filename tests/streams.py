"""Stream builders shared by the tests (test infrastructure).

Crafted lz4 / gzip images follow SURVEY.md Appendix D (the behaviour table that
was recorded with the real reference); synthetic many-block streams come from
tools/la_synth.c.
"""
import ctypes as C
import os
import struct
import zlib

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_synth = None


def synth_lib():
    global _synth
    if _synth is None:
        lib = C.CDLL(os.path.join(ROOT, "tools", "libla_synth.so"))
        lib.la_synth_lz4_frame_bound.restype = C.c_uint64
        lib.la_synth_lz4_frame_bound.argtypes = [C.c_uint32, C.c_uint32]
        lib.la_synth_lz4_stream.restype = C.c_uint64
        lib.la_synth_lz4_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.la_synth_lz4_block.restype = C.c_uint32
        lib.la_synth_lz4_block.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        _synth = lib
    return _synth


def synth_lz4_stream(seed, first_frame, nframes, blocks_per_frame=16, block_size=65536, nthreads=8,
                     want_plain=True):
    """(stream uint8 ndarray, plain uint8 ndarray or None)"""
    lib = synth_lib()
    cap = int(lib.la_synth_lz4_frame_bound(blocks_per_frame, block_size)) * nframes
    out = np.empty(cap, dtype=np.uint8)
    plain = np.empty(nframes * blocks_per_frame * block_size, dtype=np.uint8) if want_plain else None
    n = lib.la_synth_lz4_stream(seed, first_frame, nframes, blocks_per_frame, block_size, nthreads,
                                plain.ctypes.data if want_plain else None, out.ctypes.data, cap)
    assert n > 0
    return out[:n].copy(), plain


# ---------------------------------------------------------------- lz4 crafted

MAGIC = struct.pack("<I", 0x184D2204)
LEGACY = struct.pack("<I", 0x184C2102)


def lz4_desc(flg, bd, content_size=None, dict_id=None, bad_hc=False):
    d = bytes([flg, bd])
    if flg & 0x08:
        d += struct.pack("<Q", content_size if content_size is not None else 0)
    if flg & 0x01:
        d += struct.pack("<I", dict_id if dict_id is not None else 0)
    hc = (O.xxh32(d) >> 8) & 0xFF
    if bad_hc:
        hc ^= 0x55
    return d + bytes([hc])


def lz4_block(payload, stored=False, bsum=False, bad_sum=False):
    w = len(payload) | (0x80000000 if stored else 0)
    b = struct.pack("<I", w) + payload
    if bsum:
        s = O.xxh32(payload)
        if bad_sum:
            s ^= 1
        b += struct.pack("<I", s)
    return b


def lz4_frame(blocks_plain_and_payload, flg=0x64, bd=0x40, bad_content=False, **kw):
    """blocks: list of (decoded bytes, encoded block bytes)."""
    out = MAGIC + lz4_desc(flg, bd, **kw)
    plain = b""
    for p, enc in blocks_plain_and_payload:
        out += enc
        plain += p
    out += struct.pack("<I", 0)
    if flg & 0x04:
        s = O.xxh32(plain)
        if bad_content:
            s ^= 1
        out += struct.pack("<I", s)
    return out, plain


def lz4_compress_block(data):
    """Real compressor output when liblz4 is around (same image on the GPU box); else literals only."""
    try:
        l = C.CDLL("liblz4.so.1")
        l.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        buf = C.create_string_buffer(len(data) + len(data) // 255 + 64)
        n = l.LZ4_compress_default(data, buf, len(data), len(buf))
        assert n > 0
        return buf.raw[:n]
    except OSError:
        n = len(data)
        tok = bytes([min(n, 15) << 4])
        ext = b""
        if n >= 15:
            r = n - 15
            ext = b"\xff" * (r // 255) + bytes([r % 255])
        return tok + ext + data


P38 = b"The quick brown fox jumps over a dog.\n"
assert len(P38) == 38


def appendix_d_lz4_cases():
    """name -> (image, expected delivered bytes, expected rc, expected message) per SURVEY Appendix D."""
    cases = {}
    st = lambda flg=0x64, **kw: lz4_frame([(P38, lz4_block(P38, stored=True, bsum=bool(flg & 0x10)))], flg=flg, **kw)
    img, pl = st()
    cases["stored_block"] = (img, pl, 0, "")
    img, pl = lz4_frame([(P38, lz4_block(P38, stored=True))], flg=0x6D, content_size=38, dict_id=7)
    cases["content_size_dictid"] = (img, pl, 0, "")
    img, pl = lz4_frame([(P38, lz4_block(P38, stored=True))], flg=0x6C, content_size=999)
    cases["wrong_content_size"] = (img, pl, 0, "")
    img, pl = st(bad_content=True)
    cases["bad_content_sum"] = (img, pl, -30, "lz4 stream checksum error")
    img, pl = st(bad_hc=True)
    cases["bad_header_check"] = (img, b"", -30, "malformed lz4 data")
    img, pl = lz4_frame([(P38, lz4_block(P38, stored=True, bsum=True, bad_sum=True))], flg=0x74)
    cases["bad_block_sum"] = (img, b"", -30, "malformed lz4 data")
    img, pl = lz4_frame([(P38, lz4_block(P38, stored=True, bsum=True))], flg=0x74)
    cases["good_block_sum"] = (img, pl, 0, "")
    img, pl = st(flg=0x60)
    cases["no_checksums"] = (img, pl, 0, "")
    f1, p1 = st()
    skip = struct.pack("<II", 0x184D2A53, 5) + b"abcde"
    cases["frame_skip_frame"] = (f1 + skip + f1, p1 + p1, 0, "")
    cases["cut_mid_block"] = (f1[:20], b"", -30, "truncated lz4 input")
    cases["cut_before_endmark"] = (f1[:7 + 4 + 38], p1, -30, "truncated lz4 input")
    cases["cut_before_content_sum"] = (f1[:-4], p1, -30, "truncated lz4 input")
    big = bytes(70000)
    img = MAGIC + lz4_desc(0x64, 0x40) + struct.pack("<I", 70000 | 0x80000000) + big
    cases["stored_too_big"] = (img, b"", -30, "malformed lz4 data")
    img, pl = lz4_frame([(b"abc", lz4_block(b"\x30abc"))])
    cases["literals_only"] = (img, pl, 0, "")
    rle = bytes([0x1f]) + b"a" + b"\x01\x00" + b"\x0a" + b"\x50" + b"bcdef"
    img, pl = lz4_frame([(b"a" * 30 + b"bcdef", lz4_block(rle))])
    cases["overlap_rle"] = (img, pl, 0, "")
    far = bytes([0x1f]) + b"a" + b"\x09\x00" + b"\x0a" + b"\x50" + b"bcdef"
    img, _ = lz4_frame([(b"", lz4_block(far))], flg=0x60)
    cases["offset_before_start"] = (img, b"", -30, "lz4 decompression failed")
    img, _ = lz4_frame([(b"", lz4_block(b"\x00")), (P38, lz4_block(P38, stored=True))], flg=0x60)
    cases["zero_byte_block_ends_stream"] = (img, b"", 0, "")
    e, _ = lz4_frame([], flg=0x64)
    cases["empty_then_data"] = (e + f1, b"", 0, "")
    cases["data_empty_data"] = (f1 + e + f1, p1, 0, "")
    cases["legacy"] = (LEGACY + struct.pack("<I", 4) + b"\x30abc", b"abc", 0, "")
    cases["legacy_then_modern"] = (LEGACY + struct.pack("<I", 4) + b"\x30abc" + f1, b"abc" + p1, 0, "")
    cases["trailing_garbage"] = (f1 + b"garbage!!", p1, 0, "")
    return cases


def lz4_dependent_frame(chunks):
    """A block-dependent frame (FLG without bit 5) whose blocks only use in-block matches
    plus explicit references into the previous block, hand-assembled."""
    blocks = []
    prev = b""
    for i, c in enumerate(chunks):
        if i == 0 or len(prev) < 8:
            enc = lz4_compress_block(c)
            blocks.append((c, lz4_block(enc)))
        else:
            # token: 0 literals + match of 8 bytes at offset len(prev) (start of previous block), then literals
            head = prev[:8]
            seq = bytes([0x04]) + struct.pack("<H", len(prev)) + bytes([min(len(c), 15) << 4])
            n = len(c)
            if n >= 15:
                r = n - 15
                seq += b"\xff" * (r // 255) + bytes([r % 255])
            seq += c
            blocks.append((head + c, lz4_block(seq)))
        prev = blocks[-1][0]
    return lz4_frame(blocks, flg=0x44)


# ---------------------------------------------------------------- gzip crafted

def gz_member(data, level=6, name=None, comment=None, extra=None, hcrc=False, mtime=0, strategy=zlib.Z_DEFAULT_STRATEGY,
              bad_crc=False, bad_isize=False):
    flg = (4 if extra is not None else 0) | (8 if name else 0) | (16 if comment else 0) | (2 if hcrc else 0)
    h = b"\x1f\x8b\x08" + bytes([flg]) + struct.pack("<I", mtime) + b"\x00\x03"
    if extra is not None:
        h += struct.pack("<H", len(extra)) + extra
    if name:
        h += name + b"\x00"
    if comment:
        h += comment + b"\x00"
    if hcrc:
        h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = co.compress(data) + co.flush()
    crc = zlib.crc32(data) ^ (1 if bad_crc else 0)
    isz = (len(data) + (1 if bad_isize else 0)) & 0xFFFFFFFF
    return h + body + struct.pack("<II", crc, isz)
