/*
 * la_cat.c -- bsdcat-shaped driver over the public API (config 1 plumbing):
 * support_filter_all + format_empty + format_raw, open_filename with 20*512
 * byte blocks, archive_read_data_into_fd(a, 1)  (cat/bsdcat.c:74-94).
 * `-b <bytes>` first on the command line sets the block size handed to
 * archive_read_open_filename (what bsdtar's -b does, tar/read.c:203-204; the file reader
 * accepts up to 64 MiB, archive_read_open_filename.c:388-396): with bsdcat's 10 KiB blocks
 * the A-level rate is bounded by the 64 KiB read() calls underneath, not by the filter.
 * Built as the `la_cat` executable by host/Makefile.
 */
#include "../../include/la_archive.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef LA_CAT_MAIN
int main(int argc, char **argv)
{
	int rc = 0, first = 1;
	size_t block = 20 * 512;
	if (argc > 2 && strcmp(argv[1], "-b") == 0) {
		block = (size_t)strtoull(argv[2], NULL, 0);
		if (block == 0)
			block = 20 * 512;
		first = 3;
	}
	for (int i = first; i < argc || (argc == first && i == first); i++) {
		const char *fn = argc > first ? argv[i] : NULL;
		struct archive *a = archive_read_new();
		struct archive_entry *ae;
		archive_read_support_filter_all(a);
		archive_read_support_format_empty(a);
		archive_read_support_format_raw(a);
		if (archive_read_open_filename(a, fn, block) != ARCHIVE_OK) {
			fprintf(stderr, "la_cat: %s: %s\n", fn ? fn : "stdin", archive_error_string(a) ? archive_error_string(a) : "(null)");
			rc = 1;
			archive_read_free(a);
			continue;
		}
		if (archive_read_next_header(a, &ae) == ARCHIVE_OK) {
			if (archive_read_data_into_fd(a, 1) != ARCHIVE_OK) {
				fprintf(stderr, "la_cat: %s: %s\n", fn ? fn : "stdin", archive_error_string(a) ? archive_error_string(a) : "(null)");
				rc = 1;
			}
		}
		archive_read_free(a);
	}
	return rc;
}
#endif
