"""Device LZ4 compression (la_gpu_lz4_compress): the frames it writes must decode to the input
with everything that reads the format -- the oracle's filter (a port of the reference reader, with every XXH32
check), the system's liblz4 block by block, and this repository's own device decoder -- and must obey the
format's end-of-block rules.  The compressed bytes themselves are not liblz4's (an LZ4 stream is not unique)."""
import ctypes as C
import random
import struct

import numpy as np
import pytest

import oracle_lib as O
import streams as S

pytestmark = pytest.mark.gpu


def _inputs():
    rnd = random.Random(4242)
    words = [rnd.randbytes(rnd.randint(2, 11)) for _ in range(300)]
    text = b"".join(rnd.choice(words) for _ in range(120000))
    yield "empty", b""
    yield "tiny", b"abc"
    yield "twelve", b"0123456789ab"
    yield "thirteen_same", b"a" * 13
    yield "zeros_1block", bytes(65536)
    yield "zeros_ragged", bytes(3 * 65536 + 777)
    yield "random", rnd.randbytes(200000)
    yield "text", text
    yield "text_then_random", text[:300000] + rnd.randbytes(70000) + text[:5000]
    yield "period3", b"xyz" * 50000
    _, plain = S.synth_lz4_stream(7, 0, 3, blocks_per_frame=16, block_size=65536, nthreads=2)
    yield "c2_like", plain.tobytes()


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_round_trip_through_every_decoder(gpu_ctx, name, data):
    import torch
    from libarchive_amd import _native as N
    from libarchive_amd.lz4 import compress_to_frames, decode_image
    d_plain = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda() if data else torch.zeros(0, dtype=torch.uint8, device="cuda")
    for bs, bpf, flags in ((65536, 16, 3), (65536, 1, 0), (4096, 5, 1), (1000, 3, 2)):
        img = compress_to_frames(gpu_ctx, d_plain, bs, bpf, flags).cpu().numpy()
        # 1. the oracle's lz4 filter (reference reader restated, all checksums enforced)
        out, res = O.lz4_stream_decode(img, len(data) + 64)
        assert (res.rc, res.errmsg) == (0, b"") and out.tobytes() == data, (name, bs, bpf, flags)
        # 2. this repository's device decoder
        got, rc, msg, _ = decode_image(gpu_ctx, img)
        assert (rc, msg) == (0, "") and got.tobytes() == data
        # 3. frame structure + liblz4 on every block
        nb = (len(data) + bs - 1) // bs
        assert len(img) <= N.gpu_lib().la_gpu_lz4_compress_bound(len(data), bs, bpf)
        if data:
            idx = N.lz4_index(img)
            assert len(idx.blocks) == nb and len(idx.frames) == (nb + bpf - 1) // bpf
            try:
                lz = C.CDLL("liblz4.so.1")
            except OSError:
                continue
            lz.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
            raw = img.tobytes()
            for k, b in enumerate(idx.blocks):
                want = data[k * bs:(k + 1) * bs]
                pay = raw[int(b["src_off"]):int(b["src_off"]) + int(b["src_len"])]
                if int(b["flags"]) & N.LA_LZ4B_STORED:
                    assert pay == want
                    continue
                buf = C.create_string_buffer(len(want) + 1)
                assert lz.LZ4_decompress_safe(pay, buf, len(pay), len(want)) == len(want) and buf.raw[:len(want)] == want


def test_compression_ratio_is_in_liblz4s_range(gpu_ctx):
    """Not a parity claim -- a sanity bound: on compressible text the device compressor must land within
    30 % of liblz4's default level, and incompressible blocks must be stored, not inflated."""
    import torch
    from libarchive_amd.lz4 import compress_to_frames
    rnd = random.Random(1)
    words = [rnd.randbytes(rnd.randint(2, 11)) for _ in range(300)]
    text = b"".join(rnd.choice(words) for _ in range(400000))[:2 << 20]
    try:
        lz = C.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("no system liblz4")
    lz.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    ref = 0
    for i in range(0, len(text), 65536):
        blk = text[i:i + 65536]
        buf = C.create_string_buffer(len(blk) + len(blk) // 255 + 16)
        ref += lz.LZ4_compress_default(blk, buf, len(blk), len(buf))
    d = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
    mine = int(compress_to_frames(gpu_ctx, d, 65536, 16, 0).numel())
    assert mine < 1.3 * ref + 4096, (mine, ref)
    noise = torch.from_numpy(np.frombuffer(rnd.randbytes(1 << 20), dtype=np.uint8).copy()).cuda()
    assert int(compress_to_frames(gpu_ctx, noise, 65536, 16, 0).numel()) <= (1 << 20) + 16 * 4 + 15 + 64
