"""GPU parity (run with -m gpu on the MI355X box): the HIP lz4 path through the C ABI
(la_gpu_lz4_decode) against the oracle on the same inputs -- bit exact."""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import streams as S

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
MANIFEST = [e for e in json.load(open(os.path.join(GOLD, "manifest.json"))) if e["codec"] == "lz4"]


def gpu_decode(ctx, image, options=None):
    """Runs ALL THREE expand paths (LDS-window kernel + general kernel, general kernel only, the in-order
    LDS-window kernel of round 3) and BOTH parse generations (LDS-staged fused parse+checksum, and the two first-generation
    kernels) and insists that all of them agree before returning the result."""
    from libarchive_amd.lz4 import decode_image
    from libarchive_amd import _native as N
    res = []
    for opt in ((0, N.LA_LZ4_OPT_GENERAL_ONLY, N.LA_LZ4_OPT_PARSE_V1,
                 N.LA_LZ4_OPT_PARSE_V1 | N.LA_LZ4_OPT_GENERAL_ONLY,
                 N.LA_LZ4_OPT_EXPAND_INORDER) if options is None else (options,)):
        out, rc, msg, plan = decode_image(ctx, image, options=opt)
        res.append((out.tobytes(), rc, msg))
    assert all(r == res[0] for r in res), "kernel variants disagree (expand window/general/in-order x parse staged/v1)"
    return res[0]


def test_library_is_loaded_and_device_present(gpu_ctx):
    import libarchive_amd as la
    assert la.gpu_lib().la_gpu_abi_version() == 3
    assert la.gpu_lib().la_gpu_device_count() >= 1


@pytest.mark.parametrize("name", sorted(S.appendix_d_lz4_cases()))
def test_behaviour_table(gpu_ctx, name):
    img, want, rc, msg = S.appendix_d_lz4_cases()[name]
    assert gpu_decode(gpu_ctx, img) == (want, rc, msg)


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["file"] for e in MANIFEST])
def test_reference_fixtures(gpu_ctx, entry):
    import hashlib
    data = open(os.path.join(GOLD, entry["file"]), "rb").read()
    out, rc, msg = gpu_decode(gpu_ctx, data)
    assert rc == 0 and len(out) == entry["decoded_size"]
    assert hashlib.sha256(out).hexdigest() == entry["decoded_sha256"]


def test_synthetic_many_block_stream(gpu_ctx):
    img, plain = S.synth_lz4_stream(0x4C413335, 0, 64, blocks_per_frame=16, block_size=65536)
    out, rc, msg = gpu_decode(gpu_ctx, img)
    assert rc == 0 and msg == ""
    assert out == plain.tobytes()
    ref, res = O.lz4_stream_decode(img, plain.size + 16)
    assert res.rc == 0 and ref.tobytes() == out


def test_ragged_block_sizes(gpu_ctx):
    for bs in (1, 13, 64, 65, 4095, 65535):
        img, plain = S.synth_lz4_stream(77, 0, 3, blocks_per_frame=5, block_size=bs, nthreads=1)
        out, rc, msg = gpu_decode(gpu_ctx, img)
        assert (rc, msg) == (0, "") and out == plain.tobytes(), bs


def test_real_compressor_blocks_and_overlaps(gpu_ctx):
    rnd = random.Random(21)
    words = [rnd.randbytes(rnd.randint(1, 12)) for _ in range(30)]
    blocks = []
    for k in range(40):
        n = rnd.randint(1, 65536)
        kind = k % 4
        if kind == 0:
            d = b"".join(rnd.choice(words) for _ in range(n // 5 + 1))[:n]
        elif kind == 1:
            d = bytes([k]) * n                      # offset-1 RLE
        elif kind == 2:
            d = (b"ab" * n)[:n]                     # offset-2 overlap
        else:
            d = rnd.randbytes(n)                    # incompressible
        blocks.append((d, S.lz4_block(S.lz4_compress_block(d), bsum=True)))
    img, plain = S.lz4_frame(blocks, flg=0x74)
    out, rc, msg = gpu_decode(gpu_ctx, img)
    assert (rc, msg) == (0, "") and out == plain


def test_many_short_sequences_take_the_segmented_kernel(gpu_ctx):
    """> 4096 sequences in one 64 KiB block (13 107 here): more than one LDS segment of the
    window kernel; such blocks are listed on the device and expanded in segments."""
    rnd = random.Random(4)
    seqs = bytearray()
    plain = bytearray()
    while len(plain) < 65536 - 64:
        lit = rnd.randbytes(1)
        off = rnd.randint(1, min(len(plain) + 1, 65535))
        seqs += bytes([0x10]) + lit + off.to_bytes(2, "little")      # 1 literal + 4-byte match
        plain += lit
        for _ in range(4):
            plain.append(plain[-off])
    fin = rnd.randbytes(65536 - len(plain))
    seqs += bytes([0xF0]) + bytes([len(fin) - 15]) + fin if len(fin) >= 15 else bytes([len(fin) << 4]) + fin
    plain += fin
    img, pl = S.lz4_frame([(bytes(plain), S.lz4_block(bytes(seqs), bsum=True))] * 3, flg=0x74)
    ref, res = O.lz4_stream_decode(img, 1 << 20)
    assert res.rc == 0 and ref.tobytes() == pl
    assert gpu_decode(gpu_ctx, img) == (pl, 0, "")


def test_long_matches_and_chains(gpu_ctx):
    """Long matches spanning many sequences, chained back-references and period-k overlaps."""
    rnd = random.Random(12)
    blocks = []
    for k in range(12):
        base = rnd.randbytes(rnd.randint(40, 300))
        d = bytearray(base)
        while len(d) < 60000:
            mode = rnd.random()
            if mode < 0.4:
                o = rnd.randint(1, len(d)); n = rnd.randint(4, 2000)
                for _ in range(n):
                    d.append(d[-o])
            elif mode < 0.7:
                o = rnd.randint(1, 7); n = rnd.randint(4, 500)
                for _ in range(n):
                    d.append(d[-o])
            else:
                d += rnd.randbytes(rnd.randint(1, 60))
        d = bytes(d[:rnd.randint(30000, 65536)])
        blocks.append((d, S.lz4_block(S.lz4_compress_block(d), bsum=True)))
    img, plain = S.lz4_frame(blocks, flg=0x74)
    assert gpu_decode(gpu_ctx, img) == (plain, 0, "")


def test_dependent_blocks(gpu_ctx):
    chunks = [bytes(range(256)) * 3, b"hello world " * 20, b"x" * 100, b"tail" * 50]
    img, plain = S.lz4_dependent_frame(chunks)
    assert gpu_decode(gpu_ctx, img) == (plain, 0, "")


def test_error_order_in_a_long_stream(gpu_ctx):
    img, plain = S.synth_lz4_stream(5, 0, 8, blocks_per_frame=4, block_size=8192, nthreads=2)
    import libarchive_amd as la
    idx = la.lz4_index(img)
    b = idx.blocks[13]
    m = img.copy()
    m[int(b["src_off"]) + 5] ^= 0x40            # corrupt block 13: its block checksum fails first
    out, rc, msg = gpu_decode(gpu_ctx, m)
    assert (rc, msg) == (-30, "malformed lz4 data") and out == plain.tobytes()[:13 * 8192]
    ref, res = O.lz4_stream_decode(m, plain.size + 16)
    assert (ref.tobytes(), res.rc, res.errmsg.decode()) == (out, rc, msg)


def test_mutated_streams(gpu_ctx):
    rnd = random.Random(3)
    base, _ = S.synth_lz4_stream(9, 0, 2, blocks_per_frame=3, block_size=2048, nthreads=1)
    base = base.tobytes()
    for t in range(60):
        m = bytearray(base)
        for _ in range(rnd.randint(1, 3)):
            m[rnd.randrange(len(m))] = rnd.getrandbits(8)
        if rnd.random() < 0.3:
            m = m[:rnd.randrange(1, len(m))]
        m = bytes(m)
        ref, res = O.lz4_stream_decode(m, 1 << 22)
        assert gpu_decode(gpu_ctx, m) == (ref.tobytes(), res.rc, res.errmsg.decode()), t


def test_mutated_blocks_without_checksums(gpu_ctx):
    """Corrupt payloads in frames WITHOUT block checksums so that the decoder's own
    accept/reject rules decide (and nothing may be written outside a block's window)."""
    rnd = random.Random(8)
    words = [rnd.randbytes(rnd.randint(1, 9)) for _ in range(20)]
    for t in range(40):
        blocks = []
        for k in range(6):
            n = rnd.randint(20, 3000)
            d = b"".join(rnd.choice(words) for _ in range(n // 4 + 1))[:n]
            c = bytearray(S.lz4_compress_block(d))
            if k == 3:
                for _ in range(rnd.randint(1, 3)):
                    c[rnd.randrange(len(c))] = rnd.getrandbits(8)
            blocks.append((d, S.lz4_block(bytes(c))))
        img, _ = S.lz4_frame(blocks, flg=0x60)
        ref, res = O.lz4_stream_decode(img, 1 << 22)
        assert gpu_decode(gpu_ctx, img) == (ref.tobytes(), res.rc, res.errmsg.decode()), t


def test_one_stream_split_over_ranks(gpu_ctx):
    """The multi-GPU partition of bench.py on one device: ONE stream, indexed once by the product's host
    walker, cut by libarchive_amd.shard.split_stream on frame boundaries (balanced by C + U); every
    "rank" uploads and decodes ONLY its byte range through the device C ABI.  The concatenation of the
    ranges' outputs is the whole stream's output (what the oracle decodes from the whole image), and
    every range verifies its own checksums."""
    import torch
    import libarchive_amd as la
    from libarchive_amd.lz4 import Lz4DevicePlan
    from libarchive_amd.shard import slice_index, split_stream
    img, plain = S.synth_lz4_stream(77, 0, 37, blocks_per_frame=5, block_size=65536, nthreads=4)
    ref, res = O.lz4_stream_decode(img, plain.size + 16)
    assert res.rc == 0 and ref.tobytes() == plain.tobytes()
    idx = la.lz4_index(img)
    for world in (1, 2, 3, 8):
        got = []
        for lo, hi in split_stream(idx, img.size, world):
            sub, b_lo, b_hi = slice_index(idx, img.size, lo, hi)
            if hi == lo:
                continue
            d_src = torch.from_numpy(img[b_lo:b_hi].copy()).cuda()
            plan = Lz4DevicePlan(gpu_ctx, d_src, sub)
            plan.run()
            delivered, rc, msg = plan.resolve()
            sm = plan.summary()
            assert (rc, msg) == (0, "") and int(sm["n_bad_units"]) == 0 and int(sm["n_bad_frames"]) == 0
            got.append(plan.d_dst[:delivered].cpu().numpy().tobytes())
        assert b"".join(got) == plain.tobytes(), world


def test_length_extension_run_cannot_wrap_the_sum(gpu_ctx):
    """A legacy block at the size bound can carry 8.4 million 0xff extension bytes: 15 + 255 * n passes
    2^31.  liblz4 (size_t) and the oracle (long) reject the block ("lz4 decompression failed"); a 32-bit
    sum on the device must not wrap into a small or negative length and accept it."""
    import struct
    n_ext = 8421510
    for lit_run in (False, True):
        if lit_run:     # literal length 15 + 255 * n_ext
            payload = bytes([0xF0]) + b"\xff" * n_ext + b"\x07" + b"x" * 8
        else:           # match length
            payload = bytes([0x1F]) + b"A" + b"\x01\x00" + b"\xff" * n_ext + b"\x07" + b"tail!"
        assert len(payload) <= 8421520
        img = struct.pack("<I", 0x184C2102) + struct.pack("<I", len(payload)) + payload
        ref, res = O.lz4_stream_decode(img, 1 << 24)
        assert (ref.tobytes(), res.rc, res.errmsg) == (b"", -30, b"lz4 decompression failed")
        assert gpu_decode(gpu_ctx, img) == (b"", -30, "lz4 decompression failed")


def test_blocks_of_runs_long_and_medium(gpu_ctx):
    """Overlapping matches (offset < length).  Blocks of FEW LONG sequences (a 64 KiB run, long repeats) are routed to the
    wave-wide general kernel (la_dev.h, la_lz4_long_sequences); blocks of MANY MEDIUM runs stay with the LDS-window
    kernels, whose two generations copy them differently (owner lane byte by byte / period doubling in dense passes).
    Every variant must give the oracle's bytes."""
    rnd = random.Random(606)
    blocks = []
    for k in range(24):
        kind = k % 4
        if kind == 0:
            d = bytes([k]) * 65536                                   # one run
        elif kind == 1:
            d = (bytes(rnd.randbytes(rnd.randint(1, 9))) * 40000)[:65536]     # one long repeat of a short period
        elif kind == 2:
            d = b"".join(bytes([rnd.getrandbits(8)]) * rnd.randint(60, 400) for _ in range(400))[:65536]    # many medium runs
        else:
            d = b"".join(rnd.randbytes(rnd.randint(1, 5)) * rnd.randint(20, 120) + rnd.randbytes(rnd.randint(0, 30))
                         for _ in range(500))[:65536]                # medium repeats of periods 1..5 with literals between
        blocks.append((d, S.lz4_block(S.lz4_compress_block(d), bsum=True)))
    img, plain = S.lz4_frame(blocks, flg=0x74)
    ref, res = O.lz4_stream_decode(img, len(plain) + 16)
    assert res.rc == 0 and ref.tobytes() == plain
    assert gpu_decode(gpu_ctx, img) == (plain, 0, "")


def _craft_block(rnd, nseq, lit_lo, lit_hi, ml_lo, ml_hi, near=False, period=None):
    """A hand-built LZ4 block of exactly `nseq` sequences (the last one literals only) and what it decodes to: literal
    runs of lit_lo..lit_hi bytes, matches of ml_lo..ml_hi bytes at random offsets (`near`: within the last 64 bytes, so
    that sources reach into the 64 sequences in front of them; `period`: overlapping matches of that offset)."""
    out = bytearray()
    blk = bytearray()

    def lens(v):
        b = bytearray()
        while v >= 255:
            b.append(255)
            v -= 255
        b.append(v)
        return b

    for k in range(nseq):
        last = k == nseq - 1
        ll = rnd.randint(max(lit_lo, 5 if last else (1 if len(out) == 0 else lit_lo)), max(lit_hi, 12 if last else lit_hi))
        lit = bytes(rnd.randrange(256) for _ in range(ll))
        if last:
            blk.append(min(ll, 15) << 4)
            if ll >= 15:
                blk += lens(ll - 15)
            blk += lit
            out += lit
            break
        out += lit
        ml = rnd.randint(ml_lo, ml_hi)
        if period:
            off = min(period, len(out))
        elif near:
            off = rnd.randint(1, min(64, len(out)))
        else:
            off = rnd.randint(1, len(out))
        blk.append((min(ll, 15) << 4) | min(ml - 4, 15))
        if ll >= 15:
            blk += lens(ll - 15)
        blk += lit + off.to_bytes(2, "little")
        if ml - 4 >= 15:
            blk += lens(ml - 4 - 15)
        for _ in range(ml):
            out.append(out[-off])
        if len(out) > 60000:
            nseq = k + 2      # close the block with the next (literal-only) sequence
    return bytes(blk), bytes(out)


def test_group_boundaries_of_the_in_order_kernel(gpu_ctx):
    """Blocks whose sequence counts sit on and around the in-order kernel's group size (64 sequences per group, six
    literal waves, eight-group record ring): 1, 2, 63, 64, 65, 127, 128, 129, 383, 384, 385, 512, 513 sequences, with
    far, near (in-group dependencies: chains of up to 64 rounds) and overlapping sources, literal runs up to 40 bytes
    (past the 32 the literal waves prefetch) and matches up to 300 bytes (past the 64 of the straight-line copy)."""
    rnd = random.Random(20261005)
    blocks, plain = [], bytearray()
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 383, 384, 385, 512, 513):
        for kw in (dict(lit_lo=0, lit_hi=6, ml_lo=4, ml_hi=40), dict(lit_lo=0, lit_hi=3, ml_lo=4, ml_hi=12, near=True),
                   dict(lit_lo=0, lit_hi=40, ml_lo=4, ml_hi=300), dict(lit_lo=1, lit_hi=4, ml_lo=4, ml_hi=70, period=rnd.randint(1, 20))):
            payload, dec = _craft_block(rnd, n, **kw)
            if len(dec) > 65536:
                continue
            blocks.append((dec, S.lz4_block(payload, bsum=True)))
            plain += dec
    img = b""
    for i in range(0, len(blocks), 5):
        f, _ = S.lz4_frame(blocks[i:i + 5], flg=0x74)
        img += f
    ref, res = O.lz4_stream_decode(img, len(plain) + 16)
    assert res.rc == 0 and ref.tobytes() == bytes(plain)
    out, rc, msg = gpu_decode(gpu_ctx, img)
    assert (rc, msg) == (0, "") and out == bytes(plain)
