/*
 * la_archive.h -- the slice of libarchive's PUBLIC read API that sits on the
 * hot path (raw format + gzip/lz4 filters behind archive_read_data_block),
 * as exported by the repository's own minimal host (libla_host.so).
 *
 * Names, argument meaning, return codes and error behaviour are the
 * reference's (libarchive/archive.h; each prototype cites its line there) so
 * that callers and tests read like code written against libarchive itself.
 * Only what bsdcat-style consumers of this path need is present
 * (cat/bsdcat.c:74-94, libarchive/archive_read_data_into_fd.c:105-129).
 */
#ifndef LA_ARCHIVE_H
#define LA_ARCHIVE_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

/* archive.h:230-237 */
#define ARCHIVE_EOF      1
#define ARCHIVE_OK       0
#define ARCHIVE_RETRY  (-10)
#define ARCHIVE_WARN   (-20)
#define ARCHIVE_FAILED (-25)
#define ARCHIVE_FATAL  (-30)

/* archive.h:307-321 */
#define ARCHIVE_FILTER_NONE 0
#define ARCHIVE_FILTER_GZIP 1
#define ARCHIVE_FILTER_LZ4  13
#define ARCHIVE_FILTER_ZSTD 14

#define ARCHIVE_FORMAT_RAW   0x90000	/* archive.h */
#define ARCHIVE_FORMAT_EMPTY 0x60000
#define ARCHIVE_FORMAT_ZIP        0x50000	/* archive.h:361 */
#define ARCHIVE_FORMAT_TAR        0x30000	/* archive.h:353-357 */
#define ARCHIVE_FORMAT_TAR_USTAR  (ARCHIVE_FORMAT_TAR | 1)
#define ARCHIVE_FORMAT_TAR_PAX_INTERCHANGE (ARCHIVE_FORMAT_TAR | 2)
#define ARCHIVE_FORMAT_TAR_GNUTAR (ARCHIVE_FORMAT_TAR | 4)

struct archive;
struct archive_entry;

typedef ssize_t archive_read_callback(struct archive *, void *client_data, const void **buffer);	/* archive.h:241 */
typedef int     archive_open_callback(struct archive *, void *client_data);
typedef int     archive_close_callback(struct archive *, void *client_data);

struct archive *archive_read_new(void);						/* archive.h:398 */
int  archive_read_support_filter_all(struct archive *);				/* archive.h:457 */
int  archive_read_support_filter_gzip(struct archive *);			/* archive.h:466 */
int  archive_read_support_compression_gzip(struct archive *);			/* deprecated alias, gzip.c:85-92 */
int  archive_read_support_filter_lz4(struct archive *);				/* archive.h:469 */
int  archive_read_support_filter_zstd(struct archive *);			/* archive.h:482 */
int  archive_read_support_filter_none(struct archive *);
int  archive_read_support_format_raw(struct archive *);
int  archive_read_support_format_empty(struct archive *);
int  archive_read_support_format_tar(struct archive *);			/* ustar, old tar, GNU and pax names/sizes; no sparse (la_format_tar.c) */
int  archive_read_support_format_zip(struct archive *);			/* central directory on the host, deflate + CRC32 per entry on the device (la_format_zip.c) */
int  archive_read_support_format_all(struct archive *);			/* the formats of this slice: tar + zip + empty */

int  archive_read_open(struct archive *, void *client_data, archive_open_callback *,
	archive_read_callback *, archive_close_callback *);
int  archive_read_open_memory(struct archive *, const void *buff, size_t size);	/* archive_read_open_memory.c:56-59 */
int  archive_read_open_memory2(struct archive *, const void *buff, size_t size, size_t read_size);	/* :61-87 */
int  archive_read_open_filename(struct archive *, const char *filename, size_t block_size);	/* archive_read_open_filename.c:103 */

int  archive_read_next_header(struct archive *, struct archive_entry **);	/* archive_read.c:607-669 */
int  archive_read_data_block(struct archive *, const void **buff, size_t *size, int64_t *offset);	/* archive_read.c:966-982 */
ssize_t archive_read_data(struct archive *, void *, size_t);			/* archive_read.c:814-893 */
int  archive_read_data_skip(struct archive *);					/* archive_read.c:913-939 */
int  archive_read_data_into_fd(struct archive *, int fd);			/* archive_read_data_into_fd.c */
int  archive_read_close(struct archive *);
int  archive_read_free(struct archive *);

/* ---- write side: the slice the lz4 write filter needs (host/la_write_filters.c; archive.h:785-940) ---- */
struct archive *archive_write_new(void);
int  archive_write_add_filter_lz4(struct archive *);				/* archive.h:819; archive_write_add_filter_lz4.c:94 */
int  archive_write_add_filter_gzip(struct archive *);				/* archive.h:816; archive_write_add_filter_gzip.c:98 */
int  archive_write_set_format_raw(struct archive *);				/* one entry, data passed through */
int  archive_write_set_filter_option(struct archive *, const char *m, const char *o, const char *v);	/* "lz4", "block-checksum", "1" ... */
int  archive_write_open_memory(struct archive *, void *buffer, size_t buffSize, size_t *used);
int  archive_write_open_fd(struct archive *, int fd);
int  archive_write_header(struct archive *, struct archive_entry *);
ssize_t archive_write_data(struct archive *, const void *, size_t);
int  archive_write_close(struct archive *);
int  archive_write_free(struct archive *);

const char *archive_error_string(struct archive *);
int         archive_errno(struct archive *);
void        archive_set_error(struct archive *, int error_number, const char *fmt, ...)
		__attribute__((format(printf, 3, 4)));				/* archive_util.c:179 */
void        archive_clear_error(struct archive *);

int         archive_filter_count(struct archive *);				/* archive_read.c */
int         archive_filter_code(struct archive *, int);
const char *archive_filter_name(struct archive *, int);
int64_t     archive_filter_bytes(struct archive *, int);			/* archive_read.c:1164-1169 */
int         archive_format(struct archive *);
const char *archive_format_name(struct archive *);

const char *archive_entry_pathname(struct archive_entry *);
int64_t     archive_entry_mtime(struct archive_entry *);
int         archive_entry_mtime_is_set(struct archive_entry *);
int64_t     archive_entry_size(struct archive_entry *);
int         archive_entry_size_is_set(struct archive_entry *);
unsigned    archive_entry_filetype(struct archive_entry *);			/* AE_IFREG 0100000, AE_IFDIR 0040000 ... */
unsigned    archive_entry_perm(struct archive_entry *);
void        archive_entry_set_pathname(struct archive_entry *, const char *);
void        archive_entry_set_mtime(struct archive_entry *, int64_t, long);

#define ARCHIVE_ERRNO_MISC (-1)			/* archive_platform.h:213 */
#define ARCHIVE_ERRNO_FILE_FORMAT 84		/* EILSEQ-like, archive_platform.h */

#ifdef __cplusplus
}
#endif
#endif
