/*
 * la_tarwalk.c -- `bsdtar -tf`-shaped driver over the public API (BASELINE.json configs[3]):
 * support_filter_all + support_format_all, open_filename, then the
 * archive_read_next_header / archive_read_data_block loop (tar/read.c:206-400).
 * Prints one line per entry with -v, and a summary line (entries, body bytes, a sum of all
 * body bytes so that the bodies are really touched).  Built as `la_tarwalk` by host/Makefile.
 */
#include "../../include/la_archive.h"
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
	int verbose = 0, rc = 0;
	size_t block = 20 * 512;
	for (int i = 1; i < argc; i++) {
		if (strcmp(argv[i], "-v") == 0) { verbose = 1; continue; }
		if (strcmp(argv[i], "-b") == 0 && i + 1 < argc) { block = (size_t)strtoull(argv[++i], NULL, 0); continue; }
		struct archive *a = archive_read_new();
		struct archive_entry *ae;
		archive_read_support_filter_all(a);
		archive_read_support_format_all(a);
		if (archive_read_open_filename(a, argv[i], block) != ARCHIVE_OK) {
			fprintf(stderr, "la_tarwalk: %s: %s\n", argv[i], archive_error_string(a) ? archive_error_string(a) : "(null)");
			archive_read_free(a);
			rc = 1;
			continue;
		}
		uint64_t entries = 0, bytes = 0, sum = 0;
		int r;
		while ((r = archive_read_next_header(a, &ae)) == ARCHIVE_OK) {
			const void *p;
			size_t n;
			int64_t off;
			entries++;
			if (verbose)
				printf("%s\t%" PRId64 "\n", archive_entry_pathname(ae), archive_entry_size(ae));
			while ((r = archive_read_data_block(a, &p, &n, &off)) == ARCHIVE_OK) {
				const uint64_t *q = p;
				size_t k = n / 8;
				for (size_t j = 0; j < k; j++)
					sum += q[j];
				for (size_t j = k * 8; j < n; j++)
					sum += ((const unsigned char *)p)[j];
				bytes += n;
			}
			if (r != ARCHIVE_EOF)
				break;
		}
		if (r != ARCHIVE_EOF) {
			fprintf(stderr, "la_tarwalk: %s: %s\n", argv[i], archive_error_string(a) ? archive_error_string(a) : "(null)");
			rc = 1;
		}
		printf("%s: %" PRIu64 " entries, %" PRIu64 " body bytes, sum %016" PRIx64 ", format %s, filter %s\n", argv[i],
		    entries, bytes, sum, archive_format_name(a) ? archive_format_name(a) : "-", archive_filter_name(a, 0));
		archive_read_free(a);
	}
	return rc;
}
