/*
 * la_read_core.c -- minimal host for the two read filters on machines without
 * the reference tree: just enough of libarchive's read core for
 * "open -> bid -> raw format -> archive_read_data_block" (the bsdcat shape,
 * cat/bsdcat.c:74-94).
 *
 * It restates the CONTRACTS the filters rely on, not the library:
 *   - filter bidding, highest bid wins, at most 25 stages, the final
 *     "ask for one byte" verification pull      (archive_read.c:539-594)
 *   - peek/consume with a copy buffer that only comes into play when a
 *     request spans client blocks                (archive_read.c:1319-1488, :1499-1611)
 *   - a read() of 0 bytes is end of input, a negative one is sticky-fatal
 *                                               (archive_read.c:1385-1411)
 *   - raw format: one entry "data", read_data = whatever ahead(1) exposes
 *                                               (archive_read_support_format_raw.c:95-166)
 *   - memory and file clients                   (archive_read_open_memory.c:56-120,
 *                                                archive_read_open_filename.c:388-461)
 * Inside a real libarchive build none of this file is used.
 */
#include "la_read_private.h"
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <sys/stat.h>
#include <unistd.h>

#define MAX_FILTER_STAGES 25	/* archive_read.c:539 */

/* ------------------------------------------------------------------ errors */

void archive_set_error(struct archive *a, int error_number, const char *fmt, ...)
{
	va_list ap;
	a->archive_error_number = error_number;
	if (fmt == NULL) {
		a->error = NULL;
		return;
	}
	va_start(ap, fmt);
	vsnprintf(a->error_buf, sizeof(a->error_buf), fmt, ap);
	va_end(ap);
	a->error = a->error_buf;
}

void archive_clear_error(struct archive *a)
{
	a->error = NULL;
	a->error_buf[0] = 0;
	a->archive_error_number = 0;
}

const char *archive_error_string(struct archive *a) { return a->error; }
int archive_errno(struct archive *a) { return a->archive_error_number; }

/* ------------------------------------------------------------------ entry */

const char *archive_entry_pathname(struct archive_entry *e) { return e->pathname; }
int64_t archive_entry_mtime(struct archive_entry *e) { return e->mtime; }
int archive_entry_mtime_is_set(struct archive_entry *e) { return e->mtime_set; }
int64_t archive_entry_size(struct archive_entry *e) { return e->size; }
int archive_entry_size_is_set(struct archive_entry *e) { return e->size_set; }
unsigned archive_entry_filetype(struct archive_entry *e) { return e->filetype; }
unsigned archive_entry_perm(struct archive_entry *e) { return e->mode; }
void archive_entry_set_pathname(struct archive_entry *e, const char *p)
{
	snprintf(e->pathname, sizeof(e->pathname), "%s", p ? p : "");
}
void archive_entry_set_mtime(struct archive_entry *e, int64_t t, long ns)
{
	(void)ns;
	e->mtime = t;
	e->mtime_set = 1;
}

/* ------------------------------------------------------------------ object */

struct archive *archive_read_new(void)
{
	struct archive_read *a = calloc(1, sizeof(*a));
	if (a == NULL)
		return NULL;
	a->archive.state = LA_STATE_NEW;
	return &a->archive;
}

int __archive_read_register_bidder(struct archive_read *a, void *bidder_data, const char *name,
    const struct archive_read_filter_bidder_vtable *vtable)
{
	if (a->archive.state != LA_STATE_NEW) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC,
		    "INTERNAL ERROR: Function '__archive_read_register_bidder' invoked with archive structure in wrong state");
		return ARCHIVE_FATAL;
	}
	for (int i = 0; i < 16; i++) {
		struct archive_read_filter_bidder *b = &a->bidders[i];
		if (b->vtable != NULL)
			continue;
		if (vtable->bid == NULL || vtable->init == NULL) {
			archive_set_error(&a->archive, EINVAL,
			    "Internal error: no bid/init for filter bidder");
			return ARCHIVE_FATAL;
		}
		memset(b, 0, sizeof(*b));
		b->data = bidder_data;
		b->name = name;
		b->vtable = vtable;
		return ARCHIVE_OK;
	}
	archive_set_error(&a->archive, ENOMEM, "Not enough slots for filter registration");
	return ARCHIVE_FATAL;
}

int archive_read_support_filter_none(struct archive *a) { (void)a; return ARCHIVE_OK; }

int archive_read_support_filter_all(struct archive *a)
{
	/* archive_read_support_filter_all.c:40-84, restricted to the filters this host carries */
	archive_read_support_filter_gzip(a);
	archive_read_support_filter_lz4(a);
	archive_read_support_filter_zstd(a);
	archive_clear_error(a);
	return ARCHIVE_OK;
}

/* ------------------------------------------------------------------ client proxy */

static ssize_t client_read_proxy(struct archive_read_filter *self, const void **buff)
{
	struct archive_read *a = self->archive;
	return (a->client.reader)(&a->archive, a->client.data, buff);
}

static int client_close_proxy(struct archive_read_filter *self)
{
	struct archive_read *a = self->archive;
	if (a->client.closer)
		return (a->client.closer)(&a->archive, a->client.data);
	return ARCHIVE_OK;
}

static const struct archive_read_filter_vtable none_reader_vtable = {
	.read = client_read_proxy,
	.close = client_close_proxy,
};

/* ------------------------------------------------------------------ peek / consume */

/* Pull the next block from the filter's own read(); updates the client window. */
static int refill_client_block(struct archive_read_filter *f)
{
	ssize_t n = (f->vtable->read)(f, &f->client_buff);
	if (n < 0) {
		f->client_total = f->client_avail = 0;
		f->client_next = f->client_buff = NULL;
		f->fatal = 1;
		return -1;
	}
	if (n == 0) {
		f->client_total = f->client_avail = 0;
		f->client_next = f->client_buff = NULL;
		f->end_of_file = 1;
		return 0;
	}
	f->client_total = (size_t)n;
	f->client_avail = (size_t)n;
	f->client_next = (const char *)f->client_buff;
	return 1;
}

static int grow_copy_buffer(struct archive_read_filter *f, size_t min)
{
	size_t s = f->buffer_size ? f->buffer_size : min;
	while (s < min) {
		size_t t = s * 2;
		if (t <= s)
			goto nomem;
		s = t;
	}
	char *p = malloc(s);
	if (p == NULL)
		goto nomem;
	if (f->avail > 0)
		memmove(p, f->next, f->avail);
	free(f->buffer);
	f->next = f->buffer = p;
	f->buffer_size = s;
	return 0;
nomem:
	archive_set_error(&f->archive->archive, ENOMEM, "Unable to allocate copy buffer");
	f->fatal = 1;
	return -1;
}

const void *__archive_read_filter_ahead(struct archive_read_filter *f, size_t min, ssize_t *avail)
{
	if (f->fatal) {
		if (avail) *avail = ARCHIVE_FATAL;
		return NULL;
	}
	for (;;) {
		/* the copy buffer already satisfies the request */
		if (f->avail >= min && f->avail > 0) {
			if (avail) *avail = (ssize_t)f->avail;
			return f->next;
		}
		/* everything in the copy buffer still sits in the client block:
		 * hand out the client block itself (zero copy) */
		if (f->client_total >= f->client_avail + f->avail &&
		    f->client_avail + f->avail >= min) {
			f->client_avail += f->avail;
			f->client_next -= f->avail;
			f->avail = 0;
			f->next = f->buffer;
			if (avail) *avail = (ssize_t)f->client_avail;
			return f->client_next;
		}
		/* make room at the front of the copy buffer */
		if (f->next > f->buffer && f->next + min > f->buffer + f->buffer_size) {
			if (f->avail > 0)
				memmove(f->buffer, f->next, f->avail);
			f->next = f->buffer;
		}
		if (f->client_avail == 0) {
			if (f->end_of_file) {
				if (avail) *avail = (ssize_t)f->avail;
				return NULL;
			}
			int r = refill_client_block(f);
			if (r < 0) {
				if (avail) *avail = ARCHIVE_FATAL;
				return NULL;
			}
			if (r == 0) {
				if (avail) *avail = (ssize_t)f->avail;
				return NULL;
			}
			continue;
		}
		/* the request spans client blocks: gather into the copy buffer */
		if (min > f->buffer_size && grow_copy_buffer(f, min) < 0) {
			if (avail) *avail = ARCHIVE_FATAL;
			return NULL;
		}
		size_t room = (size_t)((f->buffer + f->buffer_size) - (f->next + f->avail));
		size_t want = room;
		if (want + f->avail > min)
			want = min - f->avail;
		if (want > f->client_avail)
			want = f->client_avail;
		memcpy(f->next + f->avail, f->client_next, want);
		f->client_next += want;
		f->client_avail -= want;
		f->avail += want;
	}
}

const void *__archive_read_ahead(struct archive_read *a, size_t min, ssize_t *avail)
{
	return __archive_read_filter_ahead(a->filter, min, avail);
}

static int64_t advance_file_pointer(struct archive_read_filter *f, int64_t request)
{
	int64_t done = 0;
	if (f->fatal)
		return -1;
	if (f->avail > 0) {
		size_t n = (size_t)(request < (int64_t)f->avail ? request : (int64_t)f->avail);
		f->next += n; f->avail -= n;
		request -= (int64_t)n; f->position += (int64_t)n; done += (int64_t)n;
	}
	if (f->client_avail > 0) {
		size_t n = (size_t)(request < (int64_t)f->client_avail ? request : (int64_t)f->client_avail);
		f->client_next += n; f->client_avail -= n;
		request -= (int64_t)n; f->position += (int64_t)n; done += (int64_t)n;
	}
	while (request > 0) {
		ssize_t got = (f->vtable->read)(f, &f->client_buff);
		if (got < 0) {
			f->client_buff = NULL;
			f->fatal = 1;
			return got;
		}
		if (got == 0) {
			f->client_buff = NULL;
			f->end_of_file = 1;
			return done;
		}
		if (got >= request) {
			f->client_next = (const char *)f->client_buff + request;
			f->client_avail = (size_t)(got - request);
			f->client_total = (size_t)got;
			done += request;
			f->position += request;
			return done;
		}
		f->position += got;
		done += got;
		request -= got;
	}
	return done;
}

int64_t __archive_read_filter_consume(struct archive_read_filter *f, int64_t request)
{
	if (request < 0)
		return ARCHIVE_FATAL;
	if (request == 0)
		return 0;
	int64_t skipped = advance_file_pointer(f, request);
	if (skipped == request)
		return skipped;
	if (skipped < 0)
		skipped = 0;
	archive_set_error(&f->archive->archive, ARCHIVE_ERRNO_MISC,
	    "Truncated input file (needed %jd bytes, only %jd available)",
	    (intmax_t)request, (intmax_t)skipped);
	return ARCHIVE_FATAL;
}

int64_t __archive_read_consume(struct archive_read *a, int64_t request)
{
	return __archive_read_filter_consume(a->filter, request);
}

/* ------------------------------------------------------------------ filters chain */

static int close_filters(struct archive_read *a)
{
	int r = ARCHIVE_OK;
	for (struct archive_read_filter *f = a->filter; f != NULL; f = f->upstream) {
		if (!f->closed && f->vtable != NULL) {
			int r1 = (f->vtable->close)(f);
			f->closed = 1;
			if (r1 < r)
				r = r1;
		}
		free(f->buffer);
		f->buffer = NULL;
	}
	return r;
}

void __archive_read_free_filters(struct archive_read *a)
{
	close_filters(a);
	while (a->filter != NULL) {
		struct archive_read_filter *up = a->filter->upstream;
		free(a->filter);
		a->filter = up;
	}
}

static int choose_filters(struct archive_read *a)
{
	for (int stage = 0; stage < MAX_FILTER_STAGES; stage++) {
		int best = 0;
		struct archive_read_filter_bidder *winner = NULL;
		for (int i = 0; i < 16; i++) {
			struct archive_read_filter_bidder *b = &a->bidders[i];
			if (b->vtable == NULL)
				continue;
			int bid = (b->vtable->bid)(b, a->filter);
			if (bid > best) {
				best = bid;
				winner = b;
			}
		}
		if (winner == NULL) {
			/* verify the chain by asking it for some data (this already decodes) */
			ssize_t avail;
			__archive_read_filter_ahead(a->filter, 1, &avail);
			if (avail < 0) {
				__archive_read_free_filters(a);
				return ARCHIVE_FATAL;
			}
			return ARCHIVE_OK;
		}
		struct archive_read_filter *f = calloc(1, sizeof(*f));
		if (f == NULL)
			return ARCHIVE_FATAL;
		f->bidder = winner;
		f->archive = a;
		f->upstream = a->filter;
		a->filter = f;
		if ((winner->vtable->init)(f) != ARCHIVE_OK) {
			__archive_read_free_filters(a);
			return ARCHIVE_FATAL;
		}
	}
	archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT,
	    "Input requires too many filters for decoding");
	return ARCHIVE_FATAL;
}

int __archive_read_header(struct archive_read *a, struct archive_entry *entry)
{
	if (a->filter == NULL || a->filter->vtable->read_header == NULL)
		return ARCHIVE_OK;
	return (a->filter->vtable->read_header)(a->filter, entry);
}

/* ------------------------------------------------------------------ formats: raw, empty */

struct raw_info {
	int64_t offset;
	int64_t unconsumed;
	int end_of_file;
};

static int raw_bid(struct archive_read *a, int best_bid)
{
	if (best_bid < 1 && __archive_read_ahead(a, 1, NULL) != NULL)
		return 1;
	return -1;
}

static int raw_read_header(struct archive_read *a, struct archive_entry *entry)
{
	struct raw_info *info = a->format->data;
	if (info->end_of_file)
		return ARCHIVE_EOF;
	a->archive.archive_format = ARCHIVE_FORMAT_RAW;
	a->archive.archive_format_name = "raw";
	archive_entry_set_pathname(entry, "data");
	return __archive_read_header(a, entry);
}

static int raw_read_data(struct archive_read *a, const void **buff, size_t *size, int64_t *offset)
{
	struct raw_info *info = a->format->data;
	ssize_t avail;
	if (info->unconsumed) {
		__archive_read_consume(a, info->unconsumed);
		info->unconsumed = 0;
	}
	if (info->end_of_file)
		return ARCHIVE_EOF;
	*buff = __archive_read_ahead(a, 1, &avail);
	*offset = info->offset;
	if (avail > 0) {
		*size = (size_t)avail;
		info->offset += avail;
		info->unconsumed = avail;
		return ARCHIVE_OK;
	}
	*size = 0;
	if (avail == 0) {
		info->end_of_file = 1;
		return ARCHIVE_EOF;
	}
	return (int)avail;
}

static int raw_cleanup(struct archive_read *a)
{
	free(a->format->data);
	a->format->data = NULL;
	return ARCHIVE_OK;
}

static int empty_bid(struct archive_read *a, int best_bid)
{
	(void)best_bid;
	if (__archive_read_ahead(a, 1, NULL) == NULL)
		return 1;
	return -1;
}
static int empty_read_header(struct archive_read *a, struct archive_entry *e)
{
	(void)e;
	a->archive.archive_format = ARCHIVE_FORMAT_EMPTY;
	a->archive.archive_format_name = "Empty file";
	return ARCHIVE_EOF;
}
static int empty_read_data(struct archive_read *a, const void **b, size_t *s, int64_t *o)
{
	(void)a; (void)b; (void)s; (void)o;
	return ARCHIVE_EOF;
}

int __archive_read_register_format(struct archive_read *a, struct archive_format_descriptor d)
{
	for (int i = 0; i < 4; i++) {
		if (a->formats[i].bid == d.bid)
			return ARCHIVE_OK;
		if (a->formats[i].bid == NULL) {
			a->formats[i] = d;
			return ARCHIVE_OK;
		}
	}
	return ARCHIVE_FATAL;
}

int archive_read_support_format_raw(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	struct raw_info *info = calloc(1, sizeof(*info));
	if (info == NULL) {
		archive_set_error(_a, ENOMEM, "Can't allocate raw_info data");
		return ARCHIVE_FATAL;
	}
	struct archive_format_descriptor d = { info, "raw", raw_bid, raw_read_header, raw_read_data, raw_cleanup, NULL };
	for (int i = 0; i < 4; i++)
		if (a->formats[i].bid == raw_bid) { free(info); return ARCHIVE_OK; }
	return __archive_read_register_format(a, d);
}

int archive_read_support_format_empty(struct archive *_a)
{
	struct archive_format_descriptor d = { NULL, "empty", empty_bid, empty_read_header, empty_read_data, NULL, NULL };
	return __archive_read_register_format((struct archive_read *)_a, d);
}

int archive_read_support_format_all(struct archive *_a)
{
	int r = archive_read_support_format_tar(_a);
	if (r == ARCHIVE_OK)
		r = archive_read_support_format_zip(_a);
	if (r != ARCHIVE_OK)
		return r;
	return archive_read_support_format_empty(_a);
}

static int choose_format(struct archive_read *a)
{
	int best = -1, slot = -1;
	for (int i = 0; i < 4; i++) {
		if (a->formats[i].bid == NULL)
			continue;
		a->format = &a->formats[i];
		int bid = a->formats[i].bid(a, best);
		if (bid == ARCHIVE_FATAL)
			return ARCHIVE_FATAL;
		if (a->filter->position != 0) {
			archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC, "format bidder consumed input");
			return ARCHIVE_FATAL;
		}
		if (bid > best) {
			best = bid;
			slot = i;
		}
	}
	if (slot < 0) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT, "No formats registered");
		return ARCHIVE_FATAL;
	}
	if (best < 1) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT, "Unrecognized archive format");
		return ARCHIVE_FATAL;
	}
	return slot;
}

/* ------------------------------------------------------------------ open */

static int open1(struct archive_read *a)
{
	archive_clear_error(&a->archive);
	if (a->archive.state != LA_STATE_NEW) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC, "archive_read_open: archive in wrong state");
		return ARCHIVE_FATAL;
	}
	if (a->client.reader == NULL) {
		archive_set_error(&a->archive, EINVAL, "No reader function provided to archive_read_open");
		a->archive.state = LA_STATE_FATAL;
		return ARCHIVE_FATAL;
	}
	if (a->client.opener != NULL) {
		int e = (a->client.opener)(&a->archive, a->client.data);
		if (e != 0) {
			if (a->client.closer)
				(a->client.closer)(&a->archive, a->client.data);
			return e;
		}
	}
	struct archive_read_filter *f = calloc(1, sizeof(*f));
	if (f == NULL)
		return ARCHIVE_FATAL;
	f->archive = a;
	f->vtable = &none_reader_vtable;
	f->name = "none";
	f->code = ARCHIVE_FILTER_NONE;
	a->filter = f;

	int e = choose_filters(a);
	if (e < ARCHIVE_WARN) {
		a->archive.state = LA_STATE_FATAL;
		return ARCHIVE_FATAL;
	}
	int slot = choose_format(a);
	if (slot < 0) {
		close_filters(a);
		a->archive.state = LA_STATE_FATAL;
		return ARCHIVE_FATAL;
	}
	a->format = &a->formats[slot];
	a->archive.state = LA_STATE_HEADER;
	return e;
}

int archive_read_open(struct archive *_a, void *client_data, archive_open_callback *opener,
    archive_read_callback *reader, archive_close_callback *closer)
{
	struct archive_read *a = (struct archive_read *)_a;
	a->client.opener = opener;
	a->client.reader = reader;
	a->client.closer = closer;
	a->client.data = client_data;
	return open1(a);
}

/* memory client: archive_read_open_memory.c:56-120 */
struct read_memory_data {
	const unsigned char *start, *p, *end;
	ssize_t read_size;
};

static ssize_t memory_read(struct archive *a, void *cd, const void **buff)
{
	struct read_memory_data *m = cd;
	(void)a;
	*buff = m->p;
	ssize_t n = m->end - m->p;
	if (n > m->read_size)
		n = m->read_size;
	m->p += n;
	return n;
}
static int memory_close(struct archive *a, void *cd)
{
	(void)a;
	free(cd);
	return ARCHIVE_OK;
}

int archive_read_open_memory2(struct archive *a, const void *buff, size_t size, size_t read_size)
{
	struct read_memory_data *m = calloc(1, sizeof(*m));
	if (m == NULL) {
		archive_set_error(a, ENOMEM, "No memory");
		return ARCHIVE_FATAL;
	}
	m->start = m->p = buff;
	m->end = m->start + size;
	m->read_size = (ssize_t)read_size;
	return archive_read_open(a, m, NULL, memory_read, memory_close);
}

int archive_read_open_memory(struct archive *a, const void *buff, size_t size)
{
	return archive_read_open_memory2(a, buff, size, size);
}

/* file client: archive_read_open_filename.c:388-461 (regular files and stdin) */
struct read_file_data {
	int fd;
	size_t block_size;
	void *buffer;
	char name[1024];
};

static int file_open(struct archive *a, void *cd)
{
	struct read_file_data *m = cd;
	if (m->name[0] == 0)
		m->fd = 0;
	else {
		m->fd = open(m->name, O_RDONLY | O_CLOEXEC);
		if (m->fd < 0) {
			archive_set_error(a, errno, "Failed to open '%s'", m->name);
			return ARCHIVE_FATAL;
		}
	}
	/* block size: a power of two >= 64 KiB, capped at 64 MiB (open_filename.c:388-396) */
	size_t bs = 64 * 1024;
	while (bs < m->block_size && bs < 64 * 1024 * 1024)
		bs *= 2;
	m->block_size = bs;
	m->buffer = malloc(bs);
	if (m->buffer == NULL) {
		archive_set_error(a, ENOMEM, "No memory");
		return ARCHIVE_FATAL;
	}
	return ARCHIVE_OK;
}

static ssize_t file_read(struct archive *a, void *cd, const void **buff)
{
	struct read_file_data *m = cd;
	*buff = m->buffer;
	for (;;) {
		ssize_t n = read(m->fd, m->buffer, m->block_size);
		if (n < 0) {
			if (errno == EINTR)
				continue;
			archive_set_error(a, errno, "Error reading '%s'", m->name[0] ? m->name : "stdin");
		}
		return n;
	}
}

static int file_close(struct archive *a, void *cd)
{
	struct read_file_data *m = cd;
	(void)a;
	if (m->fd > 0)
		close(m->fd);
	free(m->buffer);
	free(m);
	return ARCHIVE_OK;
}

int archive_read_open_filename(struct archive *a, const char *filename, size_t block_size)
{
	struct read_file_data *m = calloc(1, sizeof(*m));
	if (m == NULL) {
		archive_set_error(a, ENOMEM, "No memory");
		return ARCHIVE_FATAL;
	}
	m->fd = -1;
	m->block_size = block_size;
	if (filename)
		snprintf(m->name, sizeof(m->name), "%s", filename);
	return archive_read_open(a, m, file_open, file_read, file_close);
}

/* ------------------------------------------------------------------ header / data */

int archive_read_next_header(struct archive *_a, struct archive_entry **entryp)
{
	struct archive_read *a = (struct archive_read *)_a;
	*entryp = NULL;
	if (a->archive.state & (LA_STATE_FATAL | LA_STATE_CLOSED | LA_STATE_NEW)) {
		archive_set_error(_a, ARCHIVE_ERRNO_MISC, "archive_read_next_header: archive in wrong state");
		return ARCHIVE_FATAL;
	}
	if (a->archive.state == LA_STATE_EOF)
		return ARCHIVE_EOF;
	memset(&a->entry, 0, sizeof(a->entry));
	archive_clear_error(_a);
	/* archive_read.c:621-635: whatever the client left of the previous entry is skipped first */
	int r1 = ARCHIVE_OK;
	if (a->archive.state == LA_STATE_DATA) {
		r1 = archive_read_data_skip(_a);
		if (r1 == ARCHIVE_EOF)
			archive_set_error(_a, EIO, "Premature end-of-file.");
		if (r1 == ARCHIVE_EOF || r1 == ARCHIVE_FATAL) {
			a->archive.state = LA_STATE_FATAL;
			return ARCHIVE_FATAL;
		}
	}
	int r = a->format->read_header(a, &a->entry);
	switch (r) {
	case ARCHIVE_EOF:
		a->archive.state = LA_STATE_EOF;
		break;
	case ARCHIVE_OK:
	case ARCHIVE_WARN:
		a->archive.state = LA_STATE_DATA;
		break;
	case ARCHIVE_RETRY:
		break;
	default:
		a->archive.state = LA_STATE_FATAL;
		break;
	}
	a->read_data_output_offset = 0;
	a->read_data_remaining = 0;
	*entryp = &a->entry;
	return (r < r1 || r == ARCHIVE_EOF) ? r : r1;
}

/* archive_read.c:913-939 */
int archive_read_data_skip(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	int r;
	if (a->archive.state != LA_STATE_DATA) {
		archive_set_error(_a, ARCHIVE_ERRNO_MISC, "archive_read_data_skip: archive in wrong state");
		return ARCHIVE_FATAL;
	}
	if (a->format->read_data_skip != NULL)
		r = (a->format->read_data_skip)(a);
	else {
		const void *buff;
		size_t size;
		int64_t offset;
		while ((r = archive_read_data_block(_a, &buff, &size, &offset)) == ARCHIVE_OK)
			;
	}
	if (r == ARCHIVE_EOF)
		r = ARCHIVE_OK;
	if (a->archive.state != LA_STATE_FATAL)
		a->archive.state = LA_STATE_HEADER;
	return r;
}

int archive_read_data_block(struct archive *_a, const void **buff, size_t *size, int64_t *offset)
{
	struct archive_read *a = (struct archive_read *)_a;
	if (a->archive.state != LA_STATE_DATA) {
		archive_set_error(_a, ARCHIVE_ERRNO_MISC, "archive_read_data_block: archive in wrong state");
		return ARCHIVE_FATAL;
	}
	int r = a->format->read_data(a, buff, size, offset);
	if (r <= ARCHIVE_FATAL)
		a->archive.state = LA_STATE_FATAL;
	return r;
}

/* archive_read.c:814-893: copy loop on top of archive_read_data_block */
ssize_t archive_read_data(struct archive *_a, void *buff, size_t s)
{
	struct archive_read *a = (struct archive_read *)_a;
	char *dest = buff;
	size_t bytes_read = 0;
	while (s > 0) {
		if (a->read_data_remaining == 0) {
			const void *p;
			size_t n;
			int r = archive_read_data_block(_a, &p, &n, &a->read_data_offset);
			a->read_data_block = p;
			a->read_data_remaining = n;
			if (r == ARCHIVE_EOF)
				return (ssize_t)bytes_read;
			if (r < ARCHIVE_OK)
				return r;
		}
		size_t len = a->read_data_remaining < s ? a->read_data_remaining : s;
		if (len) {
			memcpy(dest, a->read_data_block, len);
			s -= len;
			a->read_data_block += len;
			a->read_data_remaining -= len;
			a->read_data_output_offset += (int64_t)len;
			a->read_data_offset += (int64_t)len;
			dest += len;
			bytes_read += len;
		}
	}
	return (ssize_t)bytes_read;
}

#define MAX_WRITE (1024 * 1024)
int archive_read_data_into_fd(struct archive *a, int fd)
{
	const void *buff;
	size_t size;
	int64_t off;
	int r;
	while ((r = archive_read_data_block(a, &buff, &size, &off)) == ARCHIVE_OK) {
		const char *p = buff;
		while (size > 0) {
			size_t w = size > MAX_WRITE ? MAX_WRITE : size;
			ssize_t n = write(fd, p, w);
			if (n < 0) {
				archive_set_error(a, errno, "Write error");
				return ARCHIVE_FATAL;
			}
			p += n;
			size -= (size_t)n;
		}
	}
	return r == ARCHIVE_EOF ? ARCHIVE_OK : r;
}

/* ------------------------------------------------------------------ introspection */

static struct archive_read_filter *get_filter(struct archive_read *a, int n)
{
	struct archive_read_filter *f = a->filter;
	if (n == -1 && f != NULL) {
		while (f->upstream != NULL)
			f = f->upstream;
		return f;
	}
	if (n < 0)
		return NULL;
	while (n > 0 && f != NULL) {
		f = f->upstream;
		--n;
	}
	return f;
}

int archive_filter_count(struct archive *_a)
{
	int n = 0;
	for (struct archive_read_filter *f = ((struct archive_read *)_a)->filter; f; f = f->upstream)
		n++;
	return n;
}
int archive_filter_code(struct archive *_a, int n)
{
	struct archive_read_filter *f = get_filter((struct archive_read *)_a, n);
	return f ? f->code : -1;
}
const char *archive_filter_name(struct archive *_a, int n)
{
	struct archive_read_filter *f = get_filter((struct archive_read *)_a, n);
	return f ? f->name : NULL;
}
int64_t archive_filter_bytes(struct archive *_a, int n)
{
	struct archive_read_filter *f = get_filter((struct archive_read *)_a, n);
	return f ? f->position : -1;
}
int archive_format(struct archive *a) { return a->archive_format; }
const char *archive_format_name(struct archive *a) { return a->archive_format_name; }

/* ------------------------------------------------------------------ close / free */

int archive_read_close(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	if (a->archive.state == LA_STATE_CLOSED)
		return ARCHIVE_OK;
	archive_clear_error(_a);
	int r = close_filters(a);
	a->archive.state = LA_STATE_CLOSED;
	return r;
}

int archive_read_free(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	int r = ARCHIVE_OK;
	if (_a == NULL)
		return ARCHIVE_OK;
	if (a->archive.state != LA_STATE_CLOSED)
		r = archive_read_close(_a);
	for (int i = 0; i < 4; i++)
		if (a->formats[i].cleanup) {
			a->format = &a->formats[i];
			a->formats[i].cleanup(a);
		}
	__archive_read_free_filters(a);
	free(a);
	return r;
}
