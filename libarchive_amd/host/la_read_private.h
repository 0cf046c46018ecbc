/*
 * la_read_private.h -- the filter-facing boundary of libarchive's read core,
 * restated for builds that do not have the reference tree (the GPU box).
 *
 * The three vtable/struct declarations below follow
 * libarchive/archive_read_private.h:43-118 field for field (names, order and
 * types), because they ARE the drop-in contract: host/la_filter_lz4.c and
 * host/la_filter_gzip.c are written against them and compile unchanged
 * against the real header when built inside libarchive (-DLA_IN_LIBARCHIVE,
 * see INTEGRATION.md).  Everything else in this file (struct archive,
 * struct archive_read) is this repository's own minimal core and only keeps
 * the members the path needs.
 */
#ifndef LA_READ_PRIVATE_H
#define LA_READ_PRIVATE_H

#ifdef LA_IN_LIBARCHIVE
/* Built inside libarchive: the real private headers define everything. */
#include "archive_platform.h"
#include "archive.h"
#include "archive_entry.h"
#include "archive_endian.h"
#include "archive_private.h"
#include "archive_read_private.h"
#else

#include "../../include/la_archive.h"
#include <stdlib.h>
#include <string.h>

struct archive_read;
struct archive_read_filter_bidder;
struct archive_read_filter;

/* archive_read_private.h:43-51 */
struct archive_read_filter_bidder_vtable {
	int (*bid)(struct archive_read_filter_bidder *, struct archive_read_filter *);
	int (*init)(struct archive_read_filter *);
	void (*free)(struct archive_read_filter_bidder *);
};

/* archive_read_private.h:68-74 */
struct archive_read_filter_bidder {
	void *data;
	const char *name;
	const struct archive_read_filter_bidder_vtable *vtable;
};

/* archive_read_private.h:76-83 */
struct archive_read_filter_vtable {
	ssize_t (*read)(struct archive_read_filter *, const void **);
	int (*close)(struct archive_read_filter *self);
	int (*read_header)(struct archive_read_filter *self, struct archive_entry *entry);
};

/* archive_read_private.h:90-118 */
struct archive_read_filter {
	int64_t position;
	struct archive_read_filter_bidder *bidder;
	struct archive_read_filter *upstream;
	struct archive_read *archive;
	const struct archive_read_filter_vtable *vtable;
	void *data;

	const char	*name;
	int		 code;
	int		 can_skip;
	int		 can_seek;

	char		*buffer;
	size_t		 buffer_size;
	char		*next;
	size_t		 avail;
	const void	*client_buff;
	size_t		 client_total;
	const char	*client_next;
	size_t		 client_avail;
	char		 end_of_file;
	char		 closed;
	char		 fatal;
};

/* ---- this repository's minimal core objects ---- */

#define LA_STATE_NEW    1
#define LA_STATE_HEADER 2
#define LA_STATE_DATA   4
#define LA_STATE_EOF    0x10
#define LA_STATE_CLOSED 0x20
#define LA_STATE_FATAL  0x8000

struct archive {
	unsigned int state;
	int          archive_format;
	const char  *archive_format_name;
	int          archive_error_number;
	char         error_buf[512];
	const char  *error;		/* NULL when no error string is set */
};

struct archive_entry {
	char    pathname[1024];
	int64_t mtime;
	int     mtime_set;
	int64_t size;			/* set by the tar walker */
	int     size_set;
	unsigned filetype;		/* AE_IF* */
	unsigned mode;			/* permission bits */
};

/* archive_entry.h:183-189 */
#define AE_IFMT   0170000u
#define AE_IFREG  0100000u
#define AE_IFLNK  0120000u
#define AE_IFCHR  0020000u
#define AE_IFBLK  0060000u
#define AE_IFDIR  0040000u
#define AE_IFIFO  0010000u

struct archive_read_client {
	archive_open_callback  *opener;
	archive_read_callback  *reader;
	archive_close_callback *closer;
	void *data;
};

struct archive_format_descriptor {
	void *data;
	const char *name;
	int (*bid)(struct archive_read *, int best_bid);
	int (*read_header)(struct archive_read *, struct archive_entry *);
	int (*read_data)(struct archive_read *, const void **, size_t *, int64_t *);
	int (*cleanup)(struct archive_read *);
	int (*read_data_skip)(struct archive_read *);	/* optional (archive_read.c:925-932) */
};

struct archive_read {
	struct archive archive;		/* must be first: filters use &self->archive->archive */
	struct archive_entry entry;
	struct archive_read_client client;
	struct archive_read_filter_bidder bidders[16];	/* archive_read_private.h:172 */
	struct archive_read_filter *filter;		/* head of the chain */
	struct archive_format_descriptor formats[4];
	struct archive_format_descriptor *format;
	/* archive_read_data() copy state (archive_read.c:814-893) */
	const char *read_data_block;
	int64_t read_data_offset, read_data_output_offset;
	size_t read_data_remaining;
};

/* archive_read_private.h:242-253 */
int __archive_read_register_bidder(struct archive_read *a, void *bidder_data, const char *name,
	const struct archive_read_filter_bidder_vtable *vtable);
int __archive_read_register_format(struct archive_read *a, struct archive_format_descriptor d);
const void *__archive_read_ahead(struct archive_read *, size_t, ssize_t *);
const void *__archive_read_filter_ahead(struct archive_read_filter *, size_t, ssize_t *);
int64_t __archive_read_consume(struct archive_read *, int64_t);
int64_t __archive_read_filter_consume(struct archive_read_filter *, int64_t);
int __archive_read_header(struct archive_read *, struct archive_entry *);
void __archive_read_free_filters(struct archive_read *);

static inline uint32_t archive_le32dec(const void *pp)	/* archive_endian.h */
{
	const unsigned char *p = (const unsigned char *)pp;
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

#endif /* !LA_IN_LIBARCHIVE */
#endif
