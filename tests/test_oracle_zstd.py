"""The zstd oracle (oracle/orc_zstd.c) against the reference's own zstd fixtures and the image's libzstd."""
import hashlib
import json
import os
import random

import pytest

import zstd_support as Z

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
MANIFEST = [e for e in json.load(open(os.path.join(GOLD, "manifest.json"))) if e["codec"] == "zstd"]


@pytest.fixture(scope="module")
def orc():
    return Z.oracle_lib()


@pytest.fixture(scope="module")
def zlib_():
    z = Z.libzstd()
    if z is None:
        pytest.skip("no libzstd.so.1 in this image")
    return z


def test_manifest_has_the_reference_fixtures():
    assert sorted(e["file"] for e in MANIFEST) == ["test_compat_zstd_1.tar.zst", "test_compat_zstd_2.tar.zst",
                                                   "test_empty.zst", "test_expand.zst"]


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["file"] for e in MANIFEST])
def test_reference_fixtures(orc, entry):
    data = open(os.path.join(GOLD, entry["file"]), "rb").read()
    assert orc.orc_zstd_bid(data, len(data)) == 32
    rc, out, msg = Z.oracle_decode(orc, data, entry["decoded_size"] + 64)
    assert (rc, msg) == (0, "") and len(out) == entry["decoded_size"]
    assert hashlib.sha256(out).hexdigest() == entry["decoded_sha256"]


def test_expand_fixture_text(orc):
    # cat/test/test_expand_zstd.c:18
    data = open(os.path.join(GOLD, "test_expand.zst"), "rb").read()
    assert Z.oracle_decode(orc, data, 100)[1] == b"contents of test_expand.zst.\n"


def test_xxh64_known_answers(orc):
    # published XXH64 values (xxHash project test vectors, seed 0 / prime seed)
    assert orc.orc_xxh64(b"", 0, 0) == 0xEF46DB3751D8E999
    assert orc.orc_xxh64(b"a", 1, 0) == 0xD24EC4F1A98C6E5B
    assert orc.orc_xxh64(b"abc", 3, 0) == 0x44BC2CF5AD770999
    s = b"Nobody inspects the spammish repetition"
    assert orc.orc_xxh64(s, len(s), 0) == 0xFBCEA83C8A378BF1


def test_bid(orc):
    # archive_read_support_filter_zstd.c:107-131
    assert orc.orc_zstd_bid(b"\x28\xb5\x2f\xfd", 4) == 32
    for k in range(16):
        assert orc.orc_zstd_bid(bytes([0x50 + k, 0x2A, 0x4D, 0x18]), 4) == 32
    assert orc.orc_zstd_bid(b"\x28\xb5\x2f", 3) == 0
    assert orc.orc_zstd_bid(b"\x04\x22\x4d\x18", 4) == 0


def test_round_trips_every_level(orc, zlib_):
    rnd = random.Random(0x5A)
    for it in range(300):
        n = rnd.choice([0, 1, 5, 100, 1000, 5000, 70000, 200000, 400000])
        d = Z.gen(rnd, n, rnd.randint(0, 4))
        img = Z.zstd_compress(zlib_, d, rnd.choice([-5, 1, 3, 5, 9, 15, 19]))
        rc, out, msg = Z.oracle_decode(orc, img, n + 16)
        assert (rc, msg) == (0, "") and out == d, (it, n)


def test_frames_back_to_back_with_skippable_frames(orc, zlib_):
    rnd = random.Random(7)
    parts, plain = [], b""
    for i in range(12):
        d = Z.gen(rnd, rnd.randint(0, 30000), rnd.randint(1, 4))
        parts.append(Z.zstd_compress(zlib_, d, 3))
        plain += d
        if i % 3 == 1:
            parts.append(Z.skippable(b"x" * (i * 7), i))
    img = b"".join(parts)
    assert Z.oracle_decode(orc, img, len(plain) + 16) == (0, plain, "")
    assert Z.zstd_decompress(zlib_, img, len(plain) + 16) == plain


def test_truncation_and_damage_verdicts(orc, zlib_):
    """Every stream libzstd refuses the oracle refuses; where both accept, the bytes agree.  (The oracle also insists on
    the exact end of every entropy stream, as libzstd 1.5 does; libzstd 1.4.8 lets a few damaged ones through.)"""
    rnd = random.Random(0x77)
    strict = 0
    for it in range(1500):
        n = rnd.choice([5, 100, 1000, 5000, 70000])
        d = Z.gen(rnd, n, rnd.randint(1, 4))
        img = bytearray(Z.zstd_compress(zlib_, d, rnd.choice([1, 3, 19])))
        trunc = rnd.random() < 0.3 and len(img) > 4
        if trunc:
            img = img[:rnd.randint(1, len(img) - 1)]
        else:
            i = rnd.randrange(len(img))
            img[i] ^= 1 << rnd.randrange(8)
        img = bytes(img)
        want = Z.zstd_decompress(zlib_, img, n + 300000)
        rc, out, msg = Z.oracle_decode(orc, img, n + 300000)
        if want is None:
            assert rc != 0, it
            if trunc:
                assert msg == "Truncated zstd input"      # zstd.c:213-217
        elif rc == 0:
            assert out == want, it
        else:
            strict += 1
    assert strict < 40
