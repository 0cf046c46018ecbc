"""Drop-in boundary checks that need the reference tree (dev container only; skipped on
the GPU box): the two filter sources compile unchanged against libarchive's REAL private
headers (-DLA_IN_LIBARCHIVE), and this repository's restatement of the filter-facing
structs (host/la_read_private.h) has the reference's layout, field for field
(libarchive/archive_read_private.h:43-118)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CFG = REF + "/contrib/android/config/linux_host.h"   # the reference's own hand-built config (archive_platform.h:42-44)

pytestmark = pytest.mark.skipif(not os.path.exists(REF + "/libarchive/archive_read_private.h"),
                                reason="reference tree not present")

REF_FLAGS = ["-D__LIBARCHIVE_BUILD", "-DHAVE_ZLIB_H", '-DPLATFORM_CONFIG_H="%s"' % CFG, "-I" + REF + "/libarchive", "-w"]

FIELDS = ["position", "bidder", "upstream", "archive", "vtable", "data", "name", "code", "can_skip", "can_seek",
          "buffer", "buffer_size", "next", "avail", "client_buff", "client_total", "client_next", "client_avail",
          "end_of_file", "closed", "fatal"]

PROBE = r"""
#include <stdio.h>
#include <stddef.h>
%s
int main(void) {
  printf("filter %%zu\n", sizeof(struct archive_read_filter));
%s
  printf("bidder %%zu %%zu %%zu %%zu\n", sizeof(struct archive_read_filter_bidder),
     offsetof(struct archive_read_filter_bidder, data), offsetof(struct archive_read_filter_bidder, name),
     offsetof(struct archive_read_filter_bidder, vtable));
  printf("bvt %%zu %%zu %%zu %%zu\n", sizeof(struct archive_read_filter_bidder_vtable),
     offsetof(struct archive_read_filter_bidder_vtable, bid), offsetof(struct archive_read_filter_bidder_vtable, init),
     offsetof(struct archive_read_filter_bidder_vtable, free));
  printf("fvt %%zu %%zu %%zu %%zu\n", sizeof(struct archive_read_filter_vtable),
     offsetof(struct archive_read_filter_vtable, read), offsetof(struct archive_read_filter_vtable, close),
     offsetof(struct archive_read_filter_vtable, read_header));
  return 0; }
"""


def _run_probe(tmp_path, name, includes, flags):
    body = "\n".join('  printf("%s %%zu\\n", offsetof(struct archive_read_filter, %s));' % (f, f) for f in FIELDS)
    src = tmp_path / (name + ".c")
    src.write_text(PROBE % (includes, body))
    exe = tmp_path / name
    subprocess.check_call(["gcc", "-o", str(exe), str(src)] + flags)
    return subprocess.check_output([str(exe)]).decode()


def test_restated_structs_have_the_reference_layout(tmp_path):
    ref = _run_probe(tmp_path, "ref", '#include "archive_platform.h"\n#include "archive.h"\n#include "archive_read_private.h"', REF_FLAGS)
    mine = _run_probe(tmp_path, "mine", '#include "%s/libarchive_amd/host/la_read_private.h"' % ROOT, ["-I" + ROOT + "/include"])
    assert mine == ref


# every file INTEGRATION.md section 1 lists as compiled inside libarchive: the three read filters, their shared bid policy,
# and the hash drop-in (which defines libarchive's own `__archive_xxhash` under this flag)
@pytest.mark.parametrize("src", ["la_filter_lz4.c", "la_filter_gzip.c", "la_filter_zstd.c", "la_bid_policy.c", "la_hash_dropin.c"])
def test_filters_compile_against_the_real_private_headers(src):
    cmd = ["gcc", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-DLA_IN_LIBARCHIVE",
           "-I" + ROOT + "/include", os.path.join(ROOT, "libarchive_amd", "host", src)] + REF_FLAGS
    subprocess.check_call(cmd)
