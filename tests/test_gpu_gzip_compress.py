"""Device gzip compression (la_gpu_gzip_compress): the members it writes must inflate to the input with
zlib (what every gzip reader, the reference's filter included, runs), with the oracle's gzip filter, and with this
repository's own device decoder; CRC32 / ISIZE trailers and the BGZF size subfields must be right.  The compressed
bytes are not zlib's (a deflate stream is not unique)."""
import gzip
import io
import random
import struct
import zlib

import numpy as np
import pytest

import la_api
import oracle_lib as O
import streams as S

pytestmark = pytest.mark.gpu


def _inputs():
    rnd = random.Random(777)
    words = [rnd.randbytes(rnd.randint(2, 11)) for _ in range(300)]
    text = b"".join(rnd.choice(words) for _ in range(120000))
    yield "one_byte", b"q"
    yield "two", b"ab"
    yield "three_same", b"aaa"
    yield "zeros", bytes(200000)
    yield "random", rnd.randbytes(150000)
    yield "text", text
    yield "period3", b"xyz" * 40000
    yield "high_bytes", bytes(range(144, 256)) * 300
    yield "long_runs", b"".join(bytes([i & 255]) * (i * 7 % 700 + 1) for i in range(600))
    _, plain = S.synth_lz4_stream(5, 0, 2, blocks_per_frame=16, block_size=65536, nthreads=2)
    yield "c2_like", plain.tobytes()


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_round_trip_through_every_inflater(gpu_ctx, name, data):
    import torch
    from libarchive_amd.gzip import compress_to_members
    d_plain = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for chunk in (49152, 1000, 4096):
        img = compress_to_members(gpu_ctx, d_plain, chunk, mtime=1234567).cpu().numpy().tobytes()
        # 1. zlib, member by member, and the size subfields
        pos, got, nm = 0, bytearray(), 0
        while pos < len(img):
            assert img[pos:pos + 4] == b"\x1f\x8b\x08\x04" and img[pos + 12:pos + 16] == b"BC\x02\x00"
            assert struct.unpack_from("<I", img, pos + 4)[0] == 1234567
            bsize = struct.unpack_from("<H", img, pos + 16)[0] + 1
            d = zlib.decompressobj(-15)
            body = d.decompress(img[pos + 18:pos + bsize])
            assert d.eof and len(d.unused_data) == 8
            crc, isize = struct.unpack("<II", d.unused_data)
            assert crc == zlib.crc32(body) and isize == len(body)
            got += body; pos += bsize; nm += 1
        assert bytes(got) == data and nm == (len(data) + chunk - 1) // chunk
        assert gzip.GzipFile(fileobj=io.BytesIO(img)).read() == data
        # 2. the oracle's gzip filter (reference reader restated)
        out, res = O.gzip_stream_decode(img, len(data) + 64)
        assert (res.rc, res.errmsg) == (0, b"") and out.tobytes() == data
        # 3. this repository's read path: bid, indexed boundaries, device inflate, CRC32 / ISIZE
        if chunk >= 4096:
            r = la_api.cat(img)
            assert r.filters[0] == (1, "gzip") and r.data == data


def test_ratio_and_stored_fallback(gpu_ctx):
    import torch
    from libarchive_amd.gzip import compress_to_members
    rnd = random.Random(2)
    words = [rnd.randbytes(rnd.randint(2, 11)) for _ in range(300)]
    text = b"".join(rnd.choice(words) for _ in range(400000))[:2 << 20]
    d = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
    mine = int(compress_to_members(gpu_ctx, d).numel())
    ref = len(zlib.compress(text, 1))
    assert mine < 1.6 * ref, (mine, ref)      # fixed Huffman codes, greedy 4096-entry table: within 60 % of zlib level 1
    noise = rnd.randbytes(1 << 20)
    dn = torch.from_numpy(np.frombuffer(noise, dtype=np.uint8).copy()).cuda()
    assert int(compress_to_members(gpu_ctx, dn).numel()) <= (1 << 20) + 22 * 31 + 64     # stored blocks, not 9/8
