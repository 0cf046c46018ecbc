"""ZIP reader through archive_read_next_header / archive_read_data_block: central directory
on the host, per-entry raw inflate + CRC32 on the device, CRC enforced.  Ground truth: the content of the
reference's own zip fixtures (tests/golden/ref_fixtures/zip, entry data recorded with Python's zipfile by
tools/make_ref_fixtures.py), Python's zipfile over archives written here, and the reference's messages
(archive_read_support_format_zip.c:3155-3196, :1180-1215)."""
import hashlib
import io
import json
import os
import random
import struct
import zipfile
import zlib

import pytest

import la_api

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures", "zip")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
ARCHIVE_FAILED = -25
AE_IFDIR, AE_IFREG = 0o040000, 0o100000


@pytest.mark.parametrize("entry", MANIFEST, ids=[e["file"] for e in MANIFEST])
def test_reference_zip_fixtures(gpu_ctx, entry):
    raw = open(os.path.join(GOLD, entry["file"]), "rb").read()
    assert hashlib.sha256(raw).hexdigest() == entry["stream_sha256"]
    want = entry["entries"]
    r = la_api.list_entries(raw)
    assert r.open_rc == 0 and r.format == 0x50000 and r.format_name.startswith("ZIP")
    bad = [i for i, e in enumerate(want) if e.get("bad_crc") or e["encrypted"] or e["method"] not in (0, 8)]
    upto = bad[0] if bad else len(want)          # la_api stops at the first entry whose data fails
    assert len(r.entries) == (upto + 1 if bad else len(want))
    for got, e in zip(r.entries[:upto], want[:upto]):
        name, size, ftype, perm, mtime, body = got
        assert name == e["name_latin1"] and size == e["size"]
        assert ftype == (AE_IFDIR if e["is_dir"] else ftype)      # directories are directories
        if "sha256" in e:
            assert len(body) == e["size"] and hashlib.sha256(body).hexdigest() == e["sha256"]
            assert zlib.crc32(body) == e["crc"]
    if bad:
        e = want[upto]
        assert r.rc == ARCHIVE_FAILED
        if e.get("bad_crc"):
            assert r.error.startswith("ZIP bad CRC: 0x") and r.error.endswith("should be 0x%x" % e["crc"])
        elif e["encrypted"]:
            assert "ncrypted" in r.error
        else:
            assert r.error.startswith("Unsupported ZIP compression method (%d: " % e["method"])
    else:
        assert r.rc == la_api.ARCHIVE_EOF and r.error is None


def _make_zip(entries, **kw):
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", **kw) as z:
        for name, data, method, level in entries:
            zi = zipfile.ZipInfo(name, date_time=(2021, 3, 4, 5, 6, 8))
            zi.compress_type = method
            zi.external_attr = (0o100640 << 16) if not name.endswith("/") else (0o040750 << 16)
            z.writestr(zi, data, compresslevel=level)
    return buf.getvalue()


def test_written_archives_many_entries(gpu_ctx):
    """Many independent deflate entries (the batch shape: one device call decodes them all), every level,
    stored entries in between, empty entries, a directory; names, sizes, modes, mtime and bodies as written."""
    rnd = random.Random(2024)
    words = [rnd.randbytes(rnd.randint(2, 12)) for _ in range(300)]
    ents = [("top/", b"", zipfile.ZIP_STORED, None)]
    for i in range(300):
        n = rnd.choice([0, 1, 17, 3000, 65536, 70001, 300000])
        data = b"".join(rnd.choice(words) for _ in range(n // 6 + 1))[:n]
        if i % 7 == 0:
            ents.append(("top/s%03d.bin" % i, data, zipfile.ZIP_STORED, None))
        else:
            ents.append(("top/d%03d.bin" % i, data, zipfile.ZIP_DEFLATED, rnd.choice([1, 6, 9])))
    img = _make_zip(ents)
    for rs in (None, 4096):
        r = la_api.list_entries(img, read_size=rs)
        assert r.rc == la_api.ARCHIVE_EOF and r.error is None and len(r.entries) == len(ents)
        for got, (name, data, method, _) in zip(r.entries, ents):
            assert got[0] == name and got[1] == len(data) and got[5] == data
            assert got[2] == (AE_IFDIR if name.endswith("/") else AE_IFREG)
            assert got[3] == (0o750 if name.endswith("/") else 0o640)
        assert r.format_name in ("ZIP 2.0 (deflation)", "ZIP 2.0 (uncompressed)", "ZIP 1.0 (uncompressed)")
    # skipping bodies costs nothing and changes nothing
    r = la_api.list_entries(img, skip_every=2)
    assert [e[0] for e in r.entries] == [e[0] for e in ents]
    assert all(g[5] == e[1] for g, e in zip(r.entries, ents) if g[5] is not None)


def test_check_values_are_enforced(gpu_ctx):
    """The reference fails an entry whose CRC32 / sizes do not match its directory record
    (zip.c:3155-3196, ARCHIVE_FAILED); so does this reader -- computed on the device for deflate entries."""
    data = b"The quick brown fox jumps over the lazy dog. " * 2000
    img = bytearray(_make_zip([("a.txt", data, zipfile.ZIP_DEFLATED, 6), ("b.txt", data[:500], zipfile.ZIP_STORED, None)]))
    cd = img.rfind(b"PK\x01\x02", 0, img.rfind(b"PK\x01\x02"))     # first central record (a.txt)
    good_crc = struct.unpack_from("<I", img, cd + 16)[0]
    assert good_crc == zlib.crc32(data)
    # wrong CRC in the directory
    m = bytearray(img); struct.pack_into("<I", m, cd + 16, good_crc ^ 0x1234)
    r = la_api.list_entries(bytes(m))
    assert r.rc == ARCHIVE_FAILED and r.error == "ZIP bad CRC: 0x%x should be 0x%x" % (good_crc, good_crc ^ 0x1234)
    # wrong uncompressed size
    m = bytearray(img); struct.pack_into("<I", m, cd + 24, len(data) - 1)
    r = la_api.list_entries(bytes(m))
    assert r.rc == ARCHIVE_FAILED and r.error.startswith("ZIP uncompressed data is wrong size")
    # damaged deflate body: the CRC (or the decode itself) reports it
    lh_data = 30 + len("a.txt")
    m = bytearray(img); m[lh_data + 200] ^= 0x40
    r = la_api.list_entries(bytes(m))
    assert r.rc in (ARCHIVE_FAILED, la_api.ARCHIVE_FATAL) and r.error.startswith(("ZIP bad CRC", "ZIP decompression failed", "ZIP uncompressed", "ZIP compressed", "Truncated ZIP"))
    # stored entry with a wrong CRC
    cd2 = img.rfind(b"PK\x01\x02")
    m = bytearray(img); struct.pack_into("<I", m, cd2 + 16, 1)
    r = la_api.list_entries(bytes(m))
    assert r.rc == ARCHIVE_FAILED and r.error.endswith("should be 0x1") and len(r.entries) == 2 and r.entries[0][5] == data
    # the untouched archive is fine, and a cut one is refused with a message
    r = la_api.list_entries(bytes(img))
    assert r.rc == la_api.ARCHIVE_EOF and [e[5] for e in r.entries] == [data, data[:500]]
    r = la_api.list_entries(bytes(img[:len(img) - 30]))
    assert r.rc == la_api.ARCHIVE_FATAL and "central directory" in r.error
