"""GPU parity of the deflate data plane (la_gpu_gzip_decode through the C ABI) against the
oracle: status, produced bytes (also on errors), consumed bytes and CRC32 -- bit exact."""
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ST_OK, ST_DATA, ST_TRUNC, ST_BAD_CRC, ST_BAD_ISIZE, ST_FULL, ST_NOTRAILER = 0, 5, 6, 7, 8, 9, 10


def gpu_inflate(ctx, bodies, caps, verify=True):
    """Runs ALL deflate paths (wave per member; lane per member in place; lane per member
    entropy decode + LDS-window expand with the in-place kernel as its fallback; the same with
    the in-order expand kernel of round 3) and insists they agree."""
    a, sa = _gpu_inflate(ctx, bodies, caps, verify, 2)
    # `consumed` only means something when the deflate stream ended (status OK / trailer verdicts)
    # (and a member that does not fit its slot is never delivered -- the filter retries it with a larger
    # slot -- so how many bytes a kernel had produced when it noticed is not part of the contract)
    norm = lambda rs: [(st, None, None, None) if st == ST_FULL else
                       (st, out, cons if st in (ST_OK, ST_BAD_CRC, ST_BAD_ISIZE, ST_NOTRAILER) else None, crc) for st, out, cons, crc in rs]
    for opt, what in ((4, "lane-per-member in-place"), (8, "two-phase (entropy decode + LDS-window expand)"),
                      (8 | 32, "two-phase (entropy decode + in-order LDS-window expand)")):
        b, sb = _gpu_inflate(ctx, bodies, caps, verify, opt)
        assert norm(a) == norm(b), "wave-per-member and %s kernels disagree" % what
        assert int(sa["n_bad_units"]) == int(sb["n_bad_units"])
        if not any(r[0] == ST_FULL for r in a):
            assert int(sa["total_out"]) == int(sb["total_out"])
    return a, sa


def _gpu_inflate(ctx, bodies, caps, verify, kernel_opt):
    """bodies: list of bytes (deflate body [+ trailer]); returns list of (status, out bytes, consumed, crc)."""
    import torch
    from libarchive_amd import _native as N
    src = b"".join(bodies)
    d_src = torch.from_numpy(np.frombuffer(src + b"\0" * 8, dtype=np.uint8).copy()).cuda()
    mem = np.zeros(len(bodies), dtype=N.GZ_MEMBER_DTYPE)
    so = do = 0
    for i, (b, c) in enumerate(zip(bodies, caps)):
        mem[i] = (so, len(b), c, do)
        so += len(b)
        do += c
    d_mem = torch.from_numpy(mem.view(np.uint8).reshape(-1).copy()).cuda()
    d_dst = torch.zeros(max(do, 16), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros(len(bodies) * 16, dtype=torch.uint8, device="cuda")
    d_sum = torch.zeros(32, dtype=torch.uint8, device="cuda")
    bt = N._GzBatchC()
    bt.d_src = d_src.data_ptr(); bt.src_bytes = len(src)
    bt.d_members = d_mem.data_ptr(); bt.n_members = len(bodies)
    bt.d_dst = d_dst.data_ptr(); bt.dst_cap = do
    bt.d_results = d_res.data_ptr(); bt.d_summary = d_sum.data_ptr()
    bt.options = (0 if verify else 1) | kernel_opt
    ctx.gzip_decode(bt)
    ctx.sync()
    res = d_res.cpu().numpy().view(N.GZ_RESULT_DTYPE)
    out = d_dst.cpu().numpy()
    r = []
    for i in range(len(bodies)):
        a = int(mem[i]["dst_off"])
        r.append((int(res[i]["status"]), out[a:a + int(res[i]["out_len"])].tobytes(), int(res[i]["consumed"]), int(res[i]["crc32"])))
    sm = d_sum.cpu().numpy().view(N.SUMMARY_DTYPE)[0]
    return r, sm


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_mid=None):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if flush_mid is None:
        return co.compress(data) + co.flush()
    h = len(data) // 2
    return co.compress(data[:h]) + co.flush(flush_mid) + co.compress(data[h:]) + co.flush()


def trailer(data):
    return (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")


def test_valid_members_all_block_types(gpu_ctx):
    rnd = random.Random(1)
    words = [rnd.randbytes(rnd.randint(1, 12)) for _ in range(40)]
    datas, bodies = [], []
    for t in range(120):
        n = rnd.choice([0, 1, 5, 100, 3000, 65536, 200000])
        kind = t % 4
        d = (b"".join(rnd.choice(words) for _ in range(n // 5 + 1))[:n] if kind == 0 else
             rnd.randbytes(n) if kind == 1 else bytes([t & 255]) * n if kind == 2 else (b"ab" * n)[:n])
        c = deflate(d, rnd.choice([0, 1, 6, 9]),
                    rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]),
                    rnd.choice([None, zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        datas.append(d)
        bodies.append(c + trailer(d) + rnd.randbytes(rnd.randint(0, 5)))
    res, sm = gpu_inflate(gpu_ctx, bodies, [len(d) for d in datas])
    for (st, out, cons, crc), d, b in zip(res, datas, bodies):
        rc, ocons, oout = O.inflate_raw(b, len(d) + 8)
        assert rc == 0 and oout == d
        assert (st, out, cons, crc) == (ST_OK, d, ocons, zlib.crc32(d) & 0xFFFFFFFF)
    assert int(sm["n_bad_units"]) == 0 and int(sm["total_out"]) == sum(len(d) for d in datas)


def test_members_with_many_short_matches(gpu_ctx):
    """64 KiB members of text-like data: thousands of short matches per member -- more than one
    LDS segment of the window kernel (4096), and for the densest ones more than the two-phase
    path's table holds (12288), which must fall back to the in-place kernel."""
    rnd = random.Random(77)
    datas = []
    for t in range(24):
        nwords = (6, 20, 200, 2000)[t % 4]
        words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz ") for _ in range(rnd.randint(2, 7))) for _ in range(nwords)]
        d = bytearray()
        while len(d) < 65536:
            d += rnd.choice(words)
        datas.append(bytes(d[:65536]))
    # three-byte matches only: the densest sequence table a deflate stream can have
    tri = bytearray(rnd.randbytes(3000))
    while len(tri) < 65536:
        a = rnd.randrange(0, len(tri) - 3)
        tri += tri[a:a + 3] + rnd.randbytes(1)
    datas.append(bytes(tri[:65536]))
    bodies = [deflate(d, 9 if i % 2 else 6) + trailer(d) for i, d in enumerate(datas)]
    res, sm = gpu_inflate(gpu_ctx, bodies, [len(d) for d in datas])
    for (st, out, cons, crc), d in zip(res, datas):
        assert (st, out, crc) == (ST_OK, d, zlib.crc32(d) & 0xFFFFFFFF)


def test_trailer_verdicts(gpu_ctx):
    d = b"The quick brown fox " * 90
    c = deflate(d)
    good = c + trailer(d)
    bad_crc = c + (zlib.crc32(d) ^ 1).to_bytes(4, "little") + len(d).to_bytes(4, "little")
    bad_isz = c + zlib.crc32(d).to_bytes(4, "little") + (len(d) + 1).to_bytes(4, "little")
    res, sm = gpu_inflate(gpu_ctx, [good, bad_crc, bad_isz, c + b"\x01\x02\x03"], [len(d)] * 4)
    assert [r[0] for r in res] == [ST_OK, ST_BAD_CRC, ST_BAD_ISIZE, ST_NOTRAILER]
    assert all(r[1] == d for r in res)
    res, _ = gpu_inflate(gpu_ctx, [bad_crc], [len(d)], verify=False)      # reference behaviour: not checked
    assert res[0][0] == ST_OK
    res, _ = gpu_inflate(gpu_ctx, [good], [len(d) - 1])
    assert res[0][0] == ST_FULL


def test_mutated_deflate_streams(gpu_ctx):
    rnd = random.Random(2)
    words = [rnd.randbytes(rnd.randint(1, 12)) for _ in range(40)]
    bodies, expect = [], []
    for t in range(400):
        n = rnd.randint(0, 6000)
        d = b"".join(rnd.choice(words) for _ in range(n // 6 + 1))[:n]
        c = bytearray(deflate(d, rnd.choice([0, 1, 6, 9]),
                              rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])))
        for _ in range(rnd.randint(1, 3)):
            if c:
                c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
        if rnd.random() < 0.3 and len(c) > 1:
            c = c[:rnd.randrange(1, len(c))]
        c = bytes(c)
        rc, cons, out = O.inflate_raw(c, 70000)
        if rc == 3:
            continue
        bodies.append(c)
        expect.append((rc, cons, out))
    res, _ = gpu_inflate(gpu_ctx, bodies, [70000] * len(bodies), verify=False)
    for i, ((st, out, cons, crc), (rc, ocons, oout)) in enumerate(zip(res, expect)):
        want = {0: (ST_OK, ST_NOTRAILER), 1: (ST_TRUNC,), 2: (ST_DATA,)}[rc]
        assert st in want, (i, st, rc)
        assert out == oout, (i, st, rc, len(out), len(oout))
        if rc == 0:
            assert cons == ocons


def test_mutated_members_in_64k_slots(gpu_ctx):
    """Mutated / truncated members whose slots are at most 64 KiB: these are the ones the two-phase
    path (entropy decode + LDS-window expand) takes, so its partial-output and error paths are what
    is compared here against the oracle and the two other kernels."""
    rnd = random.Random(11)
    words = [rnd.randbytes(rnd.randint(1, 12)) for _ in range(40)]
    bodies, caps, expect = [], [], []
    for t in range(600):
        nbytes = rnd.choice([0, 1, 100, 3000, 65536, rnd.randint(0, 65536)])
        kind = rnd.randrange(4)
        d = (b"".join(rnd.choice(words) for _ in range(nbytes // 5 + 1))[:nbytes] if kind == 0 else
             rnd.randbytes(nbytes) if kind == 1 else bytes([t & 255]) * nbytes if kind == 2 else (b"ab" * nbytes)[:nbytes])
        c = bytearray(deflate(d, rnd.choice([0, 1, 6, 9]),
                              rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])))
        if rnd.random() < 0.6:
            for _ in range(rnd.randint(1, 3)):
                if c:
                    c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
        if rnd.random() < 0.2 and len(c) > 1:
            c = c[:rnd.randrange(1, len(c))]
        cap = rnd.choice([65536, 65536, len(d), max(len(d) - 1, 0), rnd.randint(0, 65536)])
        c = bytes(c)
        bodies.append(c)
        caps.append(cap)
        expect.append(O.inflate_raw(c, cap))
    res, _ = gpu_inflate(gpu_ctx, bodies, caps, verify=False)
    for i, ((st, out, cons, crc), (rc, ocons, oout)) in enumerate(zip(res, expect)):
        want = {0: (ST_OK, ST_NOTRAILER), 1: (ST_TRUNC,), 2: (ST_DATA,), 3: (ST_FULL,)}[rc]
        assert st in want, (i, st, rc)
        if rc != 3:
            assert out == oout, (i, st, rc, len(out), len(oout))
        if rc == 0:
            assert cons == ocons


def test_many_members_batch(gpu_ctx):
    rnd = random.Random(3)
    base = rnd.randbytes(1 << 16)
    datas = []
    for t in range(600):
        a = rnd.randrange(0, 60000)
        d = (base[a:a + rnd.randint(0, 5000)] + b"xyz" * rnd.randint(0, 3000))[:65536]
        datas.append(d)
    bodies = [deflate(d, 6) + trailer(d) for d in datas]
    res, sm = gpu_inflate(gpu_ctx, bodies, [len(d) for d in datas])
    assert all(r[0] == ST_OK and r[1] == d for r, d in zip(res, datas))


# ---------------------------------------------------------------- hand-built dynamic blocks

class _Bits:
    """LSB-first bit writer (RFC 1951 3.1.1); Huffman codes go in MSB-first."""
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, nbits):
        self.acc |= (v & ((1 << nbits) - 1)) << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, nbits):
        self.put(int(format(c, "0%db" % nbits)[::-1], 2), nbits)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def _canon(lens):
    """symbol -> (code, length), canonical assignment of RFC 1951 3.2.2"""
    bl = [0] * 16
    for l in lens:
        bl[l] += 1 if l else 0
    nxt, c = [0] * 16, 0
    for b in range(1, 16):
        c = (c + bl[b - 1]) << 1
        nxt[b] = c
    out = {}
    for sy, l in enumerate(lens):
        if l:
            out[sy] = (nxt[l], l)
            nxt[l] += 1
    return out


_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEN_XB = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
_DIST_XB = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]


def _dynamic_block(ll_lens, d_lens, ops, final=True):
    """One dynamic-Huffman block with the GIVEN code lengths (286 / 30 entries).  The lengths are sent one by one
    (no repeat codes) through a flat code-length code: sixteen 4-bit words for the lengths 0..15.
    ops: ints (literal bytes) or (length symbol index, extra value, distance symbol, extra value)."""
    w = _Bits()
    w.put(1 if final else 0, 1)
    w.put(2, 2)
    nlen, ndist = 286, 30
    w.put(nlen - 257, 5)
    w.put(ndist - 1, 5)
    w.put(19 - 4, 4)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl_lens = [4 if s < 16 else 0 for s in range(19)]		# sixteen 4-bit codes: a complete code
    for s in order:
        w.put(cl_lens[s], 3)
    clc = _canon(cl_lens)
    for l in list(ll_lens) + list(d_lens):
        w.code(*clc[l])
    ll, dd = _canon(ll_lens), _canon(d_lens)
    for op in ops:
        if isinstance(op, int):
            w.code(*ll[op])
        else:
            ls, lx, ds, dx = op
            w.code(*ll[257 + ls])
            w.put(lx, _LEN_XB[ls])
            w.code(*dd[ds])
            w.put(dx, _DIST_XB[ds])
    w.code(*ll[256])
    return w


def test_hand_built_codes_of_every_length(gpu_ctx):
    """The canonical walk of the entropy decoder (per-length limits in registers, no fast table) against codes
    zlib's deflate never emits: literal/length and distance codes that use EVERY length from 1 to 15 bits, with
    symbols of 256 and up spread over the lengths (the `hib` split inside a length), a block that holds the
    end-of-block code alone (an incomplete code of one 1-bit word: legal), a lone 1-bit distance code and its
    unassigned sibling (data error), an over-subscribed code (data error).  Expected bytes from zlib's inflate."""
    rnd = random.Random(99)
    bodies, datas, expect_err = [], [], []
    for trial in range(12):
        # sixteen literal/length symbols with lengths 1..14,15,15 (Kraft sum exactly 1), assignment shuffled per trial
        pool_lit = rnd.sample(range(256), 9)
        pool_len = rnd.sample(range(29), 6)			# length symbols 257 + i
        syms = pool_lit + [256] + [257 + i for i in pool_len]
        rnd.shuffle(syms)
        ll_lens = [0] * 286
        for sy, l in zip(syms, list(range(1, 15)) + [15, 15]):
            ll_lens[sy] = l
        dsyms = rnd.sample(range(30), 16)
        d_lens = [0] * 30
        for sy, l in zip(dsyms, list(range(1, 15)) + [15, 15]):
            d_lens[sy] = l
        ops, plain = [], bytearray()
        for _ in range(rnd.randint(50, 400)):
            if plain and rnd.random() < 0.4:
                ls = rnd.choice(pool_len)
                lx = rnd.getrandbits(_LEN_XB[ls]) if _LEN_XB[ls] else 0
                length = _LEN_BASE[ls] + lx
                cands = [d for d in dsyms if _DIST_BASE[d] <= len(plain)]
                if not cands:
                    continue
                ds = rnd.choice(cands)
                dx = rnd.getrandbits(_DIST_XB[ds]) if _DIST_XB[ds] else 0
                dist = _DIST_BASE[ds] + dx
                if dist > len(plain):
                    dx, dist = 0, _DIST_BASE[ds]
                if len(plain) + length > 60000:
                    continue
                ops.append((ls, lx, ds, dx))
                for _i in range(length):
                    plain.append(plain[-dist])
            else:
                b = rnd.choice(pool_lit)
                ops.append(b)
                plain.append(b)
        body = _dynamic_block(ll_lens, d_lens, ops).done()
        assert zlib.decompressobj(-15).decompress(body) == bytes(plain)
        bodies.append(body + trailer(bytes(plain))); datas.append(bytes(plain)); expect_err.append(False)
    # a block of the end-of-block code alone, then a stored block with the payload
    only_eob = [0] * 286
    only_eob[256] = 1
    one_d = [0] * 30
    one_d[0] = 1
    w = _dynamic_block(only_eob, one_d, [], final=False)
    w.put(1, 1); w.put(0, 2)
    w.put(0, (8 - w.n) % 8)
    pay = b"stored after an empty dynamic block"
    w.put(len(pay), 16); w.put(len(pay) ^ 0xFFFF, 16)
    body = w.done() + pay
    assert zlib.decompressobj(-15).decompress(body) == pay
    bodies.append(body + trailer(pay)); datas.append(pay); expect_err.append(False)
    # one 1-bit distance code: its word works, the unassigned sibling is a data error
    ll2 = [0] * 286
    ll2[65], ll2[256], ll2[257] = 2, 2, 1
    for use_bad in (False, True):
        body = _dynamic_block(ll2, one_d, [65, (0, 0, 0, 0)]).done()
        if use_bad:
            # the same block with the OTHER distance bit: take back the end-of-block code (2 bits) behind the
            # literal, then length code, the unassigned distance word '1', end of block
            w = _dynamic_block(ll2, one_d, [65])
            total = len(w.out) * 8 + w.n - 2
            bits = int.from_bytes(bytes(w.out) + bytes([w.acc]), "little") & ((1 << total) - 1)
            w2 = _Bits(); w2.put(bits, total)
            w2.code(*_canon(ll2)[257]); w2.put(1, 1)
            w2.code(*_canon(ll2)[256])
            body = w2.done()
        bodies.append(body + trailer(b"AAAA")); datas.append(b"AAAA"); expect_err.append(use_bad)
    # over-subscribed literal/length code: three 1-bit words
    ll3 = [0] * 286
    ll3[65], ll3[66], ll3[256] = 1, 1, 1
    bodies.append(_dynamic_block(ll3, one_d, []).done() + trailer(b"")); datas.append(b""); expect_err.append(True)
    res, sm = gpu_inflate(gpu_ctx, bodies, [max(len(d), 8) + 8 for d in datas])
    for (st, out, cons, crc), d, b, bad in zip(res, datas, bodies, expect_err):
        rc, ocons, oout = O.inflate_raw(b, len(d) + 64)
        if bad:
            assert rc != 0 and st == ST_DATA and out == oout
        else:
            assert rc == 0 and oout == d
            assert (st, out, cons, crc) == (ST_OK, d, ocons, zlib.crc32(d) & 0xFFFFFFFF)
