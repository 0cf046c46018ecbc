#!/usr/bin/env python3
"""Copy the hot-path DATA fixtures of the reference's own tests into tests/golden/ref_fixtures.

Runs in the dev container only (needs /root/reference).  The inputs are the
uuencoded data files next to the reference's tests (libarchive/test, cat/test,
tar/test); they are decoded to their binary form and written with a manifest.
Expected digests of the decoded payloads are the ones measured with the real
reference -- they are what the oracle must reproduce.
"""
import binascii, hashlib, json, os, sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_fixtures")

TAR7168 = "7565705704f8f736e966783ba96277df8a37a921031a97d5a63a479d5baf1f49"
TARBIG = "b4df03c461fe9431ad097be2b6704ea993da915691cd8ecf7d7fee87ca8d595f"
EXTRACT = "7ae874a578425c31eda93582de9d10b413e18ab7378023a7bde6b82b1c37dff0"

FIXTURES = [
    # (path under reference, codec, expected decoded size, expected sha256, what the reference test asserts)
    ("libarchive/test/test_compat_lz4_1.tar.lz4.uu", "lz4", 7168, TAR7168, "test_compat_lz4.c:41-117, two concatenated frames"),
    ("libarchive/test/test_compat_lz4_2.tar.lz4.uu", "lz4", 7168, TAR7168, "trailing garbage ignored"),
    ("libarchive/test/test_compat_lz4_3.tar.lz4.uu", "lz4", 7168, TAR7168, "legacy frame"),
] + [
    ("libarchive/test/test_compat_lz4_%s.tar.lz4.uu" % v, "lz4", 4233728, TARBIG, "block size / dependence / block checksum variant " + v)
    for v in ("B4", "B4BD", "B4BDBX", "B5", "B5BD", "B6", "B6BD", "B7", "B7BD")
] + [
    ("libarchive/test/test_compat_gzip_1.tgz.uu", "gzip", 7168, TAR7168, "test_compat_gzip.c:40-94, 8 members with FNAME"),
    ("libarchive/test/test_compat_gzip_2.tgz.uu", "gzip", 7168, TAR7168, "member + trailing junk"),
    ("libarchive/test/test_read_format_raw.data.gz.uu", "gzip", 4, "b5bb9d8014a0f9b1d61e21e796d78dccdf1352f23cd32812f4850b878ae4944c", "test_read_format_raw.c:122-148 name/mtime"),
    ("cat/test/test_expand.lz4.uu", "lz4", 29, "7ff3b75ab8f584b2b9198073d0932d03dfabcdb870b21d549fdf701dff9f88d1", "cat/test/test_expand_lz4.c:10-24"),
    ("cat/test/test_expand.gz.uu", "gzip", 28, "4ac0924acf33ed4aad8411b62a0b0bd5a7ac8e16b46bd23a3ac8eaef04c997a0", "cat/test/test_expand_gz.c:10-24"),
    ("cat/test/test_empty.lz4.uu", "lz4", 0, hashlib.sha256(b"").hexdigest(), "cat/test/test_empty_lz4.c:9-23"),
    ("cat/test/test_empty.gz.uu", "gzip", 0, hashlib.sha256(b"").hexdigest(), "cat/test/test_empty_gz.c"),
    ("tar/test/test_extract.tar.lz4.uu", "lz4", 3072, EXTRACT, "tar/test/test_extract_tar_lz4.c"),
    ("tar/test/test_extract.tar.gz.uu", "gzip", 3072, EXTRACT, "tar/test/test_extract_tar_gz.c"),
    # zstd: the expected payload is what the image's libzstd (the library the reference's filter calls)
    # produces for the same bytes -- size None = computed below with ZSTD_decompress
    ("cat/test/test_expand.zst.uu", "zstd", None, None, "cat/test/test_expand_zstd.c:9-22 'contents of test_expand.zst.'"),
    ("cat/test/test_empty.zst.uu", "zstd", None, None, "cat/test/test_empty_zstd.c:9-22"),
    ("libarchive/test/test_compat_zstd_1.tar.zst.uu", "zstd", None, None, "test_compat_zstd.c:40-84: 3 frames with a skippable frame in the middle, 6 tar entries"),
    ("libarchive/test/test_compat_zstd_2.tar.zst.uu", "zstd", None, None, "test_compat_zstd.c:84-86: the same sample from pzstd"),
]


def libzstd_decompress(raw):
    import ctypes
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    z.ZSTD_isError.restype = ctypes.c_uint
    z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    buf = ctypes.create_string_buffer(1 << 24)
    n = z.ZSTD_decompress(buf, len(buf), raw, len(raw))
    assert not z.ZSTD_isError(n)
    return buf.raw[:n]


# ZIP fixtures of the reference's zip reader tests: stored / deflate entries the GPU reader takes, plus
# two with methods it must refuse per entry like the reference does.  Expected entry contents come from Python's
# zipfile over the same bytes (names, sizes, CRC32, sha256 of the body) -- the data, not any reference source text.
ZIP_FIXTURES = ["test_read_format_zip.zip", "test_read_format_zip_7075_utf8_paths.zip", "test_read_format_zip_7z_deflate.zip",
                "test_read_format_zip_comment_stored_1.zip", "test_read_format_zip_comment_stored_2.zip",
                "test_read_format_zip_high_compression.zip", "test_read_format_zip_jar.jar",
                "test_read_format_zip_length_at_end.zip", "test_read_format_zip_mac_metadata.zip",
                "test_read_format_zip_msdos.zip", "test_read_format_zip_nested.zip", "test_read_format_zip_nofiletype.zip",
                "test_read_format_zip_padded2.zip", "test_read_format_zip_symlink.zip", "test_read_format_zip_ux.zip",
                "test_read_format_zip_zip64a.zip", "test_read_format_zip_zip64b.zip",
                "test_read_format_zip_with_invalid_traditional_eocd.zip",
                "test_read_format_zip_7z_lzma.zip", "test_read_format_zip_bzip2.zipx"]


def make_zip_fixtures():
    import io, zipfile
    out = os.path.join(OUT, "zip")
    os.makedirs(out, exist_ok=True)
    manifest = []
    for name in ZIP_FIXTURES:
        rel = "libarchive/test/%s.uu" % name
        raw = uudecode(open(os.path.join(REF, rel), "r", errors="replace").read())
        open(os.path.join(out, name), "wb").write(raw)
        z = zipfile.ZipFile(io.BytesIO(raw))
        ents = []
        for i in z.infolist():
            e = {"name_latin1": i.orig_filename.encode("utf-8" if i.flag_bits & 0x800 else "cp437", "replace").decode("latin-1"),
                 "size": i.file_size, "crc": i.CRC, "method": i.compress_type, "encrypted": bool(i.flag_bits & 1),
                 "is_dir": i.is_dir()}
            if i.compress_type in (0, 8) and not (i.flag_bits & 1):
                try:
                    e["sha256"] = hashlib.sha256(z.read(i)).hexdigest()
                except zipfile.BadZipFile as err:
                    if "Bad CRC-32" in str(err):
                        e["bad_crc"] = True     # deliberately wrong check value (test_read_format_zip.c expects ARCHIVE_FAILED)
                    else:
                        # zipfile objects to something else (e.g. names that differ between the two headers): take the
                        # bytes the directory points at
                        import struct, zlib
                        nl, el = struct.unpack_from("<HH", raw, i.header_offset + 26)
                        o = i.header_offset + 30 + nl + el
                        body = raw[o:o + i.compress_size]
                        body = zlib.decompress(body, -15) if i.compress_type == 8 else body
                        assert zlib.crc32(body) == i.CRC and len(body) == i.file_size
                        e["sha256"] = hashlib.sha256(body).hexdigest()
            ents.append(e)
        manifest.append({"file": name, "source": rel, "stream_sha256": hashlib.sha256(raw).hexdigest(), "entries": ents})
        print(name, len(raw), len(ents))
    json.dump(manifest, open(os.path.join(out, "manifest.json"), "w"), indent=1)


def uudecode(text):
    out = bytearray()
    started = False
    for line in text.splitlines():
        if not started:
            if line.startswith("begin "):
                started = True
            continue
        if line.strip() == "end":
            break
        if not line:
            continue
        try:
            out += binascii.a2b_uu(line)
        except binascii.Error:
            nbytes = (((ord(line[0]) - 32) & 63) * 4 + 5) // 3
            out += binascii.a2b_uu(line[:nbytes])
    return bytes(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = []
    for rel, codec, size, sha, note in FIXTURES:
        raw = uudecode(open(os.path.join(REF, rel), "r", errors="replace").read())
        name = os.path.basename(rel)[:-3]
        open(os.path.join(OUT, name), "wb").write(raw)
        if codec == "zstd" and size is None:
            plain = libzstd_decompress(raw)
            size, sha = len(plain), hashlib.sha256(plain).hexdigest()
        manifest.append({"file": name, "source": rel, "codec": codec, "decoded_size": size,
                         "decoded_sha256": sha, "pins": note, "stream_sha256": hashlib.sha256(raw).hexdigest()})
        print(name, len(raw))
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    make_zip_fixtures()


if __name__ == "__main__":
    sys.exit(main())
