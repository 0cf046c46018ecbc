"""CPU-only: the plain-C host side END TO END -- read core, bidding, both filters (window
gathering, two-window pipelining, stream-order error resolution, gzip carry buffer and
retry hints), walkers -- driven through the reference's API shape exactly like
tests/test_gpu_filters.py, but linked against tests/mock_gpu (a CPU stand-in for the device
C ABI built from the oracle; test infrastructure, never part of the product).  The very
same test functions run on the GPU box against the real device library."""
import ctypes as C
import os
import subprocess

import pytest

import la_api

HERE = os.path.dirname(os.path.abspath(__file__))
MOCK_DIR = os.path.join(HERE, "mock_gpu")


@pytest.fixture(scope="module")
def gpu_ctx():
    """Same name as the GPU fixture on purpose: the imported tests ask for it.  Here it
    builds the mock-linked host library and routes the API harness to it."""
    subprocess.check_call(["make", "-s", "-C", MOCK_DIR])
    lib = C.CDLL(os.path.join(MOCK_DIR, "libla_host_mock.so"))
    la_api.use_library(lib)
    yield None
    la_api.use_library(None)


# the test functions themselves (their module-level `gpu` mark stays behind in that module)
from test_gpu_bid_policy import (  # noqa: E402,F401
    test_gzip_large_single_member_is_declined_many_members_are_taken,
    test_lz4_single_frame_with_content_checksum_is_declined,
    test_zstd_one_large_frame_is_declined_small_frames_are_taken,
)
from test_gpu_filters import (  # noqa: E402,F401
    test_reference_fixtures_through_the_api,
    test_gzip_entry_metadata_from_fixture,
    test_lz4_behaviour_table_through_the_api,
    test_lz4_small_reader_blocks_and_read_data,
    test_lz4_multiple_batches,
    test_lz4_frame_larger_than_the_window,
    test_lz4_window_grows_for_large_blocks,
    test_lz4_legacy_frame_across_windows,
    test_lz4_dependent_frame_across_windows,
    test_lz4_file_reader,
    test_gzip_behaviour_table_through_the_api,
    test_gzip_metadata_snapshot,
    test_gzip_strict_mode_rejects_bad_crc,
    test_gzip_mutated_streams,
    test_gzip_members_with_unusual_xfl_os_bytes,
    test_gzip_bgzf_indexed_members_through_the_filter,
    test_gzip_indexed_members_copy_ahead_across_many_windows,
    test_gzip_single_member_across_windows,
    test_lz4_window_bounded_by_decoded_bytes,
    test_gzip_bid_only_indexed_switch,
    test_gzip_window_bounded_by_decoded_bytes,
)
from test_gpu_tar import (  # noqa: E402,F401
    test_reference_tar_fixtures_list_like_the_reference_tests,
    test_extract_fixtures_contents,
    test_tarfile_written_archives_walk_entry_by_entry,
    test_c4_shape_many_equal_entries,
    test_damaged_tar_streams_fail_like_the_reference,
    test_old_style_tar_and_number_forms,
    test_gnu_and_pax_archives_with_long_names,
    test_archives_written_by_the_system_tar,
)


from test_gpu_zstd import (  # noqa: E402,F401
    test_zstd_reference_fixtures_through_the_api,
    test_zstd_compat_tars_list_like_the_reference_test,
    test_zstd_every_level_and_shape,
    test_zstd_many_frames_with_skippable_frames_and_small_reads,
    test_zstd_stream_that_starts_with_a_skippable_frame,
    test_zstd_truncated_and_damaged_streams,
    test_zstd_garbage_behind_a_frame,
    test_zstd_frame_whose_blocks_claim_more_than_the_window_budget,
    test_zstd_one_frame_larger_than_the_gather_limit,
    test_zstd_frame_of_compressed_blocks_beyond_the_window_budget,
)

from test_gpu_zip import (  # noqa: E402,F401
    test_reference_zip_fixtures,
    test_written_archives_many_entries,
    test_check_values_are_enforced,
)


def test_write_filter_host_logic_on_the_mock(gpu_ctx, monkeypatch):
    """The lz4 write filter's host side (windows, options, frame hand-off, empty stream, client errors) against the
    mock device (which stores every block): what it writes reads back through the read path."""
    import ctypes as C
    import random
    import test_gpu_lz4_write as W
    mock = C.CDLL(os.path.join(MOCK_DIR, "libla_host_mock.so"))
    monkeypatch.setattr(W, "_lib", lambda: W._lib_setup(mock))
    monkeypatch.setenv("LA_GPU_WRITE_WINDOW_MIB", "1")
    rnd = random.Random(31)
    for size in (0, 1, 65536, (1 << 20) + 7, 3 * (1 << 20) + 999):
        data = rnd.randbytes(size)
        for opts in ((), (("block-checksum", "1"),), (("stream-checksum", None), ("block-size", "5"))):
            rc, img = W.write_lz4(data, opts, rnd.choice([None, 4097]))
            assert rc == 0 and la_api.cat(img).data == data
    W.test_write_filter_options_and_errors(None)
    W.test_gzip_write_filter_round_trips(None, monkeypatch)


def test_mock_library_is_not_the_product(gpu_ctx):
    """Guard: the product library must not resolve to the mock, and vice versa."""
    import libarchive_amd as la
    prod = la.host_lib()
    assert "mock" not in getattr(prod, "_name", "")
    assert la_api._lib() is not prod


def test_lz4_window_pipeline_fuzz(gpu_ctx, monkeypatch):
    """Many 1 MiB windows per stream (two in flight inside the filter), frames crossing window
    borders, mutations and truncations anywhere: bytes / return code / error string must be the
    reference's.  Only feasible in bulk with the CPU mock."""
    import random
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(2024)
    for t in range(80):
        nfr = rnd.randint(3, 14)
        img, plain = S.synth_lz4_stream(100 + t, 0, nfr, blocks_per_frame=rnd.choice([1, 4, 7]),
                                        block_size=rnd.choice([4096, 30000, 65536]), nthreads=2)
        m = bytearray(img.tobytes())
        how = rnd.randrange(4)
        if how == 1:
            m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
        elif how == 2:
            m = m[:rnd.randrange(1, len(m))]
        elif how == 3:
            m += bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 9)))
        m = bytes(m)
        out, res = O.lz4_stream_decode(m, 1 << 26)
        want = (out.tobytes(), res.rc, res.errmsg.decode())
        for rs in (None, 65536, 1000):
            got = la_api.as_reference_tuple(la_api.cat(m, read_size=rs))
            assert got == want, (t, how, rs)


def test_gzip_window_fuzz(gpu_ctx, monkeypatch):
    """Multi-member gzip streams several windows long (1 MiB windows), members of all sizes up to
    a few hundred KiB, header variants, mutations / truncations / junk: bytes, return code and
    error string must be the reference's (64 KiB delivery rule included)."""
    import random
    import zlib
    import oracle_lib as O
    import streams as S
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    rnd = random.Random(4242)
    words = [rnd.randbytes(rnd.randint(1, 10)) for _ in range(200)]
    for t in range(80):
        parts = []
        for k in range(rnd.randint(2, 30)):
            n = rnd.choice([0, 1, 500, 20000, 65536, 70000, rnd.randint(0, 300000)])
            kind = rnd.randrange(3)
            d = (b"".join(rnd.choice(words) for _ in range(n // 5 + 1))[:n] if kind == 0 else
                 rnd.randbytes(n) if kind == 1 else bytes([k]) * n)
            parts.append(S.gz_member(d, level=rnd.choice([0, 1, 6, 9]), name=b"n%d" % k if rnd.random() < 0.3 else None))
        m = bytearray(b"".join(parts))
        how = rnd.randrange(5)
        if how == 1:
            m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
        elif how == 2:
            m = m[:rnd.randrange(1, len(m))]
        elif how == 3:
            m += rnd.randbytes(rnd.randint(1, 20))
        elif how == 4 and len(parts) > 2:
            cut = len(parts[0]) + len(parts[1]) // 2
            m[cut] ^= 0x04
        m = bytes(m)
        out, res = O.gzip_stream_decode(m, 1 << 27)
        want = (out.tobytes(), res.rc, res.errmsg.decode())
        for rs in (None, 4096):
            got = la_api.as_reference_tuple(la_api.cat(m, read_size=rs))
            assert got == want, (t, how, rs, len(got[0]), len(want[0]), got[1:], want[1:])



def test_dependent_frame_really_used_the_carried_history(gpu_ctx, monkeypatch):
    """The imported dependent-frame test passes trivially if the filter widened its window to hold
    the whole frame; the mock counts the blocks it decoded against a carried history."""
    lib = C.CDLL(os.path.join(MOCK_DIR, "libla_host_mock.so"))
    lib.la_gpu_mock_hist_blocks.restype = C.c_ulong
    before = lib.la_gpu_mock_hist_blocks()
    test_lz4_dependent_frame_across_windows(gpu_ctx, monkeypatch)
    assert lib.la_gpu_mock_hist_blocks() - before >= 3 * 4 * 5


# ---- CPU-only additions (host logic that needs no device): refusal paths with lowered limits ----

def _gz_member(data, level=1, isize=None):
    import zlib, struct
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + body +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, (len(data) if isize is None else isize) & 0xFFFFFFFF))


def test_gzip_member_beyond_the_span_limit_is_refused_by_name(gpu_ctx, monkeypatch):
    """The 4 GiB guard of the member table (la_gzip_index.c: 32-bit span) with the limit lowered to 200 000 bytes:
    members in front of the oversized one are delivered, then the explicit message -- never a wrapped span."""
    import random
    rnd = random.Random(11)
    ok_plain = rnd.randbytes(50_000)
    big_plain = rnd.randbytes(400_000)           # incompressible: the member's span is > 200 000 bytes
    img = _gz_member(ok_plain) + _gz_member(big_plain)
    monkeypatch.setenv("LA_GZ_TEST_SPAN_LIMIT", "200000")
    r = la_api.cat(img)
    assert r.rc == la_api.ARCHIVE_FATAL and r.error == "gzip member too large for the GPU data plane (4 GiB limit)"
    assert ok_plain.startswith(r.data) and len(r.data) <= len(ok_plain)      # whole 64 KiB blocks in front of the error
    monkeypatch.delenv("LA_GZ_TEST_SPAN_LIMIT")
    r = la_api.cat(img)
    assert r.rc == la_api.ARCHIVE_EOF and r.data == ok_plain + big_plain


def test_gzip_slot_that_cannot_grow_any_further_is_refused_by_name(gpu_ctx, monkeypatch):
    """The 2 GiB guard of an output slot (la_filter_gzip.c) with the limit lowered to 256 KiB: a member whose ISIZE
    field claims 10 bytes but holds 1 MiB makes the filter retry with doubled slots until the limit, then stop with
    the explicit message instead of retrying for ever or delivering a wrapped slot."""
    plain = bytes(range(256)) * 4096             # 1 MiB, compresses well: the span is small, the output is not
    img = _gz_member(plain, level=6, isize=10)
    monkeypatch.setenv("LA_GZ_TEST_SLOT_LIMIT", "262144")
    r = la_api.cat(img)
    assert r.rc == la_api.ARCHIVE_FATAL and r.error == "gzip member too large for the GPU data plane (4 GiB limit)"
    monkeypatch.delenv("LA_GZ_TEST_SLOT_LIMIT")
    r = la_api.cat(img)
    assert r.rc == la_api.ARCHIVE_EOF and r.data == plain


def test_zstd_skippable_frames_larger_than_the_window_are_passed_over(gpu_ctx, monkeypatch):
    """Skippable frames that alone fill (and pass) a gather window are dropped and the window is reused -- the stream
    is not refused as "frame too large" (zstd.c:196-260 skips them in constant memory)."""
    import zstd_support as Z
    z = Z.libzstd()
    if z is None:
        pytest.skip("no libzstd.so.1 in this image")
    a, b = b"first frame " * 1000, b"second frame " * 1000
    skip = Z.skippable(b"\0" * (3 << 20), 5)     # 3 MiB of skippable payload, window and stage limit are 1 MiB
    img = Z.zstd_compress(z, a, 3) + skip + skip + Z.zstd_compress(z, b, 3) + skip
    monkeypatch.setenv("LA_GPU_BATCH_MIB", "1")
    monkeypatch.setenv("LA_GPU_MAX_BATCH_MIB", "1")
    r = la_api.cat(img, read_size=65536)
    assert r.rc == la_api.ARCHIVE_EOF and r.error is None and r.data == a + b
