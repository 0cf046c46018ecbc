"""tar.gz / tar.lz4 through the archive_read_next_header / archive_read_data_block loop (SURVEY §8f-2,
BASELINE.json configs[3] shape): the ustar walker of libarchive_amd/host/la_format_tar.c on top of the
device filters.

Pins: the reference's own fixtures with what its tests assert on them (entry names, format code, filter
code: libarchive/test/test_compat_gzip.c:40-94, test_compat_lz4.c:41-117, tar/test/test_extract_tar_gz.c,
tar/test/test_extract_tar_lz4.c) and the error strings of archive_read_support_format_tar.c (cited per case).
Archives beyond the fixtures are written by Python's tarfile in USTAR_FORMAT and listed by it as the check.

The same functions run on the CPU mock of the device ABI (tests/test_host_filters_mock.py imports them)."""
import io
import os
import random
import tarfile

import pytest

import la_api
import streams as S

pytestmark = pytest.mark.gpu

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
ARCHIVE_EOF, ARCHIVE_OK, ARCHIVE_RETRY, ARCHIVE_FATAL = 1, 0, -10, -30
TAR_USTAR, TAR_OLD = 0x30001, 0x30000
AE_IFREG, AE_IFDIR, AE_IFLNK = 0o100000, 0o040000, 0o120000

SIX = ["f1", "f2", "f3", "d1/f1", "d1/f2", "d1/f3"]            # test_compat_gzip.c:43, test_compat_lz4.c:88
BIG3 = ["xfile", "README", "NEWS"]                              # test_compat_lz4.c:89
REF_CASES = [("test_compat_gzip_1.tgz", SIX, (1, "gzip")), ("test_compat_gzip_2.tgz", SIX, (1, "gzip")),
             ("test_compat_lz4_1.tar.lz4", SIX, (13, "lz4")), ("test_compat_lz4_2.tar.lz4", SIX, (13, "lz4")),
             ("test_compat_lz4_3.tar.lz4", SIX, (13, "lz4"))] + \
            [("test_compat_lz4_%s.tar.lz4" % v, BIG3, (13, "lz4"))
             for v in ("B4", "B5", "B6", "B7", "B4BD", "B5BD", "B6BD", "B7BD", "B4BDBX")]


@pytest.mark.parametrize("case", REF_CASES, ids=[c[0] for c in REF_CASES])
def test_reference_tar_fixtures_list_like_the_reference_tests(gpu_ctx, case):
    name, names, filt = case
    r = la_api.list_entries(None, filename=os.path.join(FIX, name), block_size=200)   # the tests open with 200
    assert r.open_rc == ARCHIVE_OK
    assert [e[0] for e in r.entries] == names
    assert r.rc == ARCHIVE_EOF and r.error is None
    assert r.filters[0] == filt
    assert r.format == TAR_USTAR
    for e in r.entries:
        assert e[2] == AE_IFREG and len(e[5]) == e[1]
    if names is SIX:
        assert [e[5] for e in r.entries] == [b"f1\n", b"f2\n", b"f3\n"] * 2


def test_extract_fixtures_contents(gpu_ctx):
    """tar/test/test_extract_tar_gz.c:17-25 and test_extract_tar_lz4.c: file1 / file2 and their text."""
    for name in ("test_extract.tar.gz", "test_extract.tar.lz4"):
        r = la_api.list_entries(open(os.path.join(FIX, name), "rb").read())
        assert [(e[0], e[5]) for e in r.entries] == [("file1", b"contents of file1.\n"), ("file2", b"contents of file2.\n")]
        assert r.rc == ARCHIVE_EOF


def _make_tar(rnd, n_entries, sizes=None, fmt=tarfile.USTAR_FORMAT):
    words = [rnd.randbytes(rnd.randint(2, 9)) for _ in range(500)]
    bio = io.BytesIO()
    want = []
    with tarfile.open(fileobj=bio, mode="w", format=fmt) as t:
        for i in range(n_entries):
            kind = rnd.random()
            if kind < 0.08:
                ti = tarfile.TarInfo("dir%d/" % i)
                ti.type = tarfile.DIRTYPE
                ti.mode = 0o755
                ti.mtime = rnd.randrange(1 << 31)
                t.addfile(ti)
                want.append(("dir%d/" % i, 0, AE_IFDIR, 0o755, ti.mtime, b""))
                continue
            if kind < 0.12:
                ti = tarfile.TarInfo("link%d" % i)
                ti.type = tarfile.SYMTYPE
                ti.linkname = "f1"
                ti.mode = 0o777
                t.addfile(ti)
                want.append(("link%d" % i, 0, AE_IFLNK, 0o777, 0, b""))
                continue
            size = rnd.choice(sizes) if sizes else rnd.choice([0, 1, 511, 512, 513, 4096, 70000, rnd.randrange(200000)])
            body = b"".join(rnd.choice(words) for _ in range(size // 5 + 1))[:size]
            name = "f%d" % i
            if kind > 0.85:
                name = "/".join("p%d" % (i + k) * 6 for k in range(6)) + "/" + "n" * 40     # needs the ustar prefix field
            ti = tarfile.TarInfo(name)
            ti.size = len(body)
            ti.mode = rnd.choice([0o644, 0o600, 0o755])
            ti.mtime = rnd.randrange(1 << 33)
            t.addfile(ti, io.BytesIO(body))
            want.append((name, len(body), AE_IFREG, ti.mode, ti.mtime, body))
    return bio.getvalue(), want


def _gz_members(data, chunk=65536, level=6):
    return b"".join(S.gz_member(data[o:o + chunk], level=level) for o in range(0, max(len(data), 1), chunk))


def _lz4_frames(data, block=65536, per_frame=16):
    out = b""
    for o in range(0, max(len(data), 1), block * per_frame):
        part = data[o:o + block * per_frame]
        blocks = [(part[b:b + block], S.lz4_block(S.lz4_compress_block(part[b:b + block]), bsum=True))
                  for b in range(0, len(part), block)]
        out += S.lz4_frame(blocks, flg=0x74)[0]
    return out


@pytest.mark.parametrize("codec", ["gz", "lz4"])
def test_tarfile_written_archives_walk_entry_by_entry(gpu_ctx, codec, monkeypatch):
    """The C4 shape in small: many entries behind many independent 64 KiB members / blocks; names through the
    prefix field, directories, symlinks, sizes around the 512-byte record; bodies read, or skipped by next_header."""
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(4242)
    tar, want = _make_tar(rnd, 120)
    image = _gz_members(tar) if codec == "gz" else _lz4_frames(tar)
    for read_size, skip_every in ((None, 0), (4096, 0), (None, 3), (65536, 2)):
        r = la_api.list_entries(image, read_size=read_size, skip_every=skip_every)
        assert r.rc == ARCHIVE_EOF and r.error is None, (r.rc, r.error)
        assert r.format == TAR_USTAR and r.format_name == "POSIX ustar format"
        assert len(r.entries) == len(want)
        for i, (got, w) in enumerate(zip(r.entries, want), 1):
            assert got[:5] == w[:5], (i, got[:5], w[:5])
            if skip_every and i % skip_every == 0:
                assert got[5] is None
            else:
                assert got[5] == w[5], i


def test_c4_shape_many_equal_entries(gpu_ctx):
    """configs[3] in miniature: N x 256 KiB entries in a tar, 64 KiB gzip members; every body comes back."""
    rnd = random.Random(7)
    tar, want = _make_tar(rnd, 24, sizes=[262144])
    r = la_api.list_entries(_gz_members(tar, level=1))
    assert r.rc == ARCHIVE_EOF
    assert [(e[0], e[1], e[5]) for e in r.entries] == [(w[0], w[1], w[5]) for w in want]


def _first_regular_offset(tar):
    """offset of the first header, past the first entry, of a regular file with a body of at least two records"""
    o = 0
    while o < len(tar):
        h = tar[o:o + 512]
        size = int(h[124:135].rstrip(b"\0 ") or b"0", 8)
        if o > 0 and h[156:157] in (b"0", b"\0") and size >= 1024:
            return o, size
        o += 512 + (size + 511) // 512 * 512
    raise AssertionError("no such entry")


def test_damaged_tar_streams_fail_like_the_reference(gpu_ctx):
    rnd = random.Random(99)
    tar, want = _make_tar(rnd, 20, sizes=[3000, 5000, 70000])
    o, size = _first_regular_offset(tar)

    # cut in the middle of a body (archive_read_support_format_tar.c:644-649)
    r = la_api.list_entries(_gz_members(tar[:o + 512 + 700]))
    assert r.rc == ARCHIVE_FATAL and r.error == "Truncated tar archive detected while reading data"
    assert r.entries[-1][5] == tar[o + 512:o + 512 + 700]

    # cut inside a header record (:769-775)
    r = la_api.list_entries(_lz4_frames(tar[:o + 100]))
    assert r.rc == ARCHIVE_FATAL and r.error == "Truncated tar archive detected while reading next header"

    # the stream ends at a record boundary without an end mark: a clean end (:757-768)
    r = la_api.list_entries(_gz_members(tar[:o]))
    assert r.rc == ARCHIVE_EOF and r.error is None

    # a body cut short and then skipped: the consume behind next_header fails (archive_read.c:1499-1521, :621-635)
    r = la_api.list_entries(_gz_members(tar[:o + 512 + 700]), read_bodies=False)
    assert r.rc == ARCHIVE_FATAL and r.error.startswith("Truncated input file (needed ")

    # header checksum (:798-809): ARCHIVE_RETRY with the message, the next call goes on with the following record
    bad = bytearray(tar)
    bad[o + 20] ^= 0x01
    r = la_api.list_entries(_gz_members(bytes(bad)))
    assert r.rc == ARCHIVE_RETRY and r.error == "Damaged tar archive (bad header checksum)"

    # one zero record ends the archive even when more follows (:778-795)
    r = la_api.list_entries(_gz_members(tar[:o] + bytes(512) + tar[o:]))
    assert r.rc == ARCHIVE_EOF and len(r.entries) == sum(1 for _ in _headers_before(tar, o))

    # a special header with nothing behind it (:757-765)
    bio = io.BytesIO()
    with tarfile.open(fileobj=bio, mode="w", format=tarfile.GNU_FORMAT) as t:
        ti = tarfile.TarInfo("y" * 300)
        ti.size = 3
        t.addfile(ti, io.BytesIO(b"abc"))
    g = bio.getvalue()
    assert g[156:157] == b"L"
    r = la_api.list_entries(_gz_members(g[:1024]))          # 'L' header + its body, then the end of the stream
    assert r.rc == ARCHIVE_FATAL and r.error == "Damaged tar archive (end-of-archive within a sequence of headers)"
    r = la_api.list_entries(_gz_members(g[:700]))           # the long name itself is cut (:1267-1274)
    assert r.rc == ARCHIVE_FATAL and r.error == "Truncated archive detected while reading metadata"


def _headers_before(tar, limit):
    o = 0
    while o < limit:
        h = tar[o:o + 512]
        size = int(h[124:135].rstrip(b"\0 ") or b"0", 8)
        if h[156:157] not in (b"0", b"\0"):
            size = 0
        yield o
        o += 512 + (size + 511) // 512 * 512


def test_old_style_tar_and_number_forms(gpu_ctx):
    """A pre-POSIX header (no magic) is 'tar (non-POSIX)' (:907-911); a base-256 size field (:3454-3494) and a
    size padded with blanks (:339-365) are read like the octal form."""
    def header(name, size_field, magic=b"ustar\x0000", typeflag=b"0"):
        h = bytearray(512)
        h[0:len(name)] = name
        h[100:108] = b"0000644\0"
        h[108:116] = b"0000000\0"
        h[116:124] = b"0000000\0"
        h[124:136] = size_field
        h[136:148] = b"00000000000\0"
        h[148:156] = b"        "
        h[156:157] = typeflag
        h[257:257 + len(magic)] = magic
        h[148:156] = b"%06o\0 " % sum(h)
        return bytes(h)
    body = b"0123456789" * 70
    pad = bytes(-len(body) % 512)
    sizes = [b"%011o\0" % len(body), b"      %o \0" % len(body) + b"\0", b"\x80" + len(body).to_bytes(11, "big")]
    tar = b"".join(header(b"n%d" % i, s[:12].ljust(12, b"\0")) + body + pad for i, s in enumerate(sizes)) + bytes(1024)
    r = la_api.list_entries(_gz_members(tar))
    assert r.rc == ARCHIVE_EOF, r.error
    assert [(e[0], e[1], e[5]) for e in r.entries] == [("n%d" % i, len(body), body) for i in range(3)]
    old = header(b"old", sizes[0], magic=b"") + body + pad + bytes(1024)
    r = la_api.list_entries(_lz4_frames(old))
    assert r.rc == ARCHIVE_EOF and r.format == TAR_OLD and r.format_name == "tar (non-POSIX)"
    assert r.entries[0][5] == body


TAR_PAX, TAR_GNU = 0x30002, 0x30004


@pytest.mark.parametrize("fmt", ["gnu", "pax"])
def test_gnu_and_pax_archives_with_long_names(gpu_ctx, fmt, monkeypatch):
    """What GNU tar and bsdtar / Python write by default: GNU headers with 'L' long names
    (archive_read_support_format_tar.c:2927-3023, :1206-1221) and pax 'x' records for path / size / mtime
    (:1846-2100), in front of ordinary entries; listing and bodies as Python's tarfile reads them back."""
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(31)
    pyfmt = tarfile.GNU_FORMAT if fmt == "gnu" else tarfile.PAX_FORMAT
    bio = io.BytesIO()
    with tarfile.open(fileobj=bio, mode="w", format=pyfmt) as t:
        for i in range(40):
            name = "f%d" % i
            if i % 3 == 0:
                name = "/".join("directory-%d-%d" % (i, k) for k in range(rnd.randint(8, 40))) + "/leaf-%d.bin" % i
            body = rnd.randbytes(rnd.choice([0, 5, 512, 700, 66000]))
            ti = tarfile.TarInfo(name)
            ti.size = len(body)
            ti.mtime = 1700000000 + i + (0.25 if fmt == "pax" and i % 4 == 0 else 0)
            ti.mode = 0o640
            t.addfile(ti, io.BytesIO(body))
        d = tarfile.TarInfo("z" * 150 + "/")
        d.type = tarfile.DIRTYPE
        t.addfile(d)
    tar = bio.getvalue()
    with tarfile.open(fileobj=io.BytesIO(tar)) as t:
        want = [(m.name + ("/" if m.isdir() else ""), m.size if m.isfile() else 0, int(m.mtime),
                 t.extractfile(m).read() if m.isfile() else b"") for m in t.getmembers()]
    assert any(len(w[0]) > 256 for w in want)
    for image in (_gz_members(tar), _lz4_frames(tar)):
        for skip_every in (0, 2):
            r = la_api.list_entries(image, read_size=rnd.choice([None, 4096]), skip_every=skip_every)
            assert r.rc == ARCHIVE_EOF and r.error is None, (r.rc, r.error)
            assert r.format == (TAR_GNU if fmt == "gnu" else TAR_PAX)
            assert [(e[0], e[1], e[4]) for e in r.entries] == [(w[0], w[1], w[2]) for w in want]
            for i, (e, w) in enumerate(zip(r.entries, want), 1):
                if not (skip_every and i % skip_every == 0):
                    assert e[5] == w[3]


def test_archives_written_by_the_system_tar(gpu_ctx, tmp_path):
    """Interop: what `tar -czf` of this image writes (GNU format, one gzip member) and its --format=ustar /
    --format=posix variants, listed entry by entry; names, sizes and bodies as the files on disk."""
    import shutil
    import subprocess
    if shutil.which("tar") is None:
        pytest.skip("no tar(1) in this image")
    rnd = random.Random(8)
    root = tmp_path / "tree"
    (root / "sub" / ("deep-" * 30)).mkdir(parents=True)
    files = {}
    for i, rel in enumerate(["a.txt", "sub/b.bin", "sub/" + "deep-" * 30 + "/" + "n" * 120, "empty"]):
        body = b"" if rel == "empty" else rnd.randbytes(rnd.choice([10, 5000, 80000]))
        (root / rel).write_bytes(body)
        files["tree/" + rel] = body
    for fmt in ("gnu", "ustar", "posix"):
        out = tmp_path / ("t_%s.tar.gz" % fmt)
        r = subprocess.run(["tar", "--format=" + fmt, "-czf", str(out), "-C", str(tmp_path), "tree"], capture_output=True)
        if r.returncode != 0:
            assert fmt == "ustar", r.stderr      # the 120-byte leaf under a long directory may not fit ustar
            continue
        res = la_api.list_entries(None, filename=str(out), block_size=65536)
        assert res.rc == ARCHIVE_EOF and res.error is None, (fmt, res.rc, res.error)
        got = {e[0]: e[5] for e in res.entries if e[2] == AE_IFREG}
        assert got == files, (fmt, sorted(got), sorted(files))
        assert sum(1 for e in res.entries if e[2] == AE_IFDIR) == 3
