/*
 * la_cat.c -- bsdcat-shaped driver over the public API (config 1 plumbing):
 * support_filter_all + format_empty + format_raw, open_filename with 20*512
 * byte blocks, archive_read_data_into_fd(a, 1)  (cat/bsdcat.c:74-94).
 * Built as the `la_cat` executable by host/Makefile.
 */
#include "../../include/la_archive.h"
#include <stdio.h>

#ifdef LA_CAT_MAIN
int main(int argc, char **argv)
{
	int rc = 0;
	for (int i = 1; i < argc || (argc == 1 && i == 1); i++) {
		const char *fn = argc > 1 ? argv[i] : NULL;
		struct archive *a = archive_read_new();
		struct archive_entry *ae;
		archive_read_support_filter_all(a);
		archive_read_support_format_empty(a);
		archive_read_support_format_raw(a);
		if (archive_read_open_filename(a, fn, 20 * 512) != ARCHIVE_OK) {
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
