/*
 * la_write_filters.c -- the lz4 and gzip WRITE filters on the device data plane, with the
 * small slice of libarchive's write side they need to stand alone.
 *
 * The filter keeps the reference's write-filter vtable and registration
 * (libarchive/archive_write_private.h:46-63 `struct archive_write_filter` {options, open, write,
 * flush, close, free, data, name, code}; archive_write_add_filter_lz4.c:94-149: name "lz4", code
 * ARCHIVE_FILTER_LZ4, defaults stream-checksum on / block-checksum off / block-size 7, options
 * :154-201) so that inside libarchive (-DLA_IN_LIBARCHIVE is not wired for this file yet) it is
 * the same kind of source-level swap as the read filters.  What differs by design:
 *   - write() only gathers input into a pinned window; a full window (LA_GPU_WRITE_WINDOW_MIB,
 *     default 64) goes to la_gpu_lz4_compress() in ONE call and the frames come back in one copy;
 *   - the stream is a sequence of frames of sixteen 64 KiB blocks, not one frame: a frame's content
 *     checksum is one serial XXH32 chain, sixteen-block frames keep thousands of chains in flight
 *     on the device (every lz4 reader, the reference's included, reads concatenated frames:
 *     archive_read_support_filter_lz4.c:328-364).  "block-size" 4..7 is accepted, blocks are 64 KiB;
 *   - "block-dependence" is refused (the device compresses independent blocks);
 *     "compression-level" 1..9 is accepted and means the one level the device has.
 *
 * The gzip filter (archive_write_add_filter_gzip.c:98-137 registration: name "gzip", code ARCHIVE_FILTER_GZIP,
 * options "compression-level" and "timestamp" :142-167) works the same way on la_gpu_gzip_compress(): the stream
 * is a sequence of members of at most 48 KiB of input each, every one with the BGZF-compatible size subfield, so
 * that the read side indexes them without searching (every gzip reader reads concatenated members:
 * archive_read_support_filter_gzip.c:340-365).  "compression-level" 0..9 is accepted and means the one level the
 * device has (fixed Huffman codes; chunks that do not shrink are stored).
 *
 * The write core below is the minimum the filters need outside libarchive: archive_write_new,
 * _add_filter_lz4, _set_format_raw (one entry, data passed through: archive_write_set_format_raw.c),
 * _set_filter_option, _open_memory / _open_fd, _header, _data, _close, _free.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "la_read_private.h"
#include "../../include/la_gpu.h"
#include "../../include/la_host.h"

struct archive_write_filter {	/* archive_write_private.h:46-63 */
	int64_t bytes_written;
	struct archive *archive;
	struct archive_write_filter *next_filter;
	int (*options)(struct archive_write_filter *, const char *key, const char *value);
	int (*open)(struct archive_write_filter *);
	int (*write)(struct archive_write_filter *, const void *, size_t);
	int (*flush)(struct archive_write_filter *);
	int (*close)(struct archive_write_filter *);
	int (*free)(struct archive_write_filter *);
	void *data;
	const char *name;
	int code;
	int bytes_per_block;
	int bytes_in_last_block;
	int state;
};

struct archive_write {
	struct archive archive;		/* first: the error helpers of la_read_core.c work on it */
	struct archive_write_filter *filter_first, *filter_last;
	/* client */
	uint8_t *mem; size_t mem_cap, *mem_used;
	int fd;
	int opened, format_raw, entries, closed;
};

/* ------------------------------------------------------------------ minimal write core */

static int client_write(struct archive_write_filter *f, const void *buf, size_t len)
{
	struct archive_write *a = (struct archive_write *)f->archive;
	if (a->mem) {
		if (*a->mem_used + len > a->mem_cap) {
			archive_set_error(&a->archive, ENOMEM, "Buffer exhausted");	/* archive_write_open_memory.c:81-85 */
			return ARCHIVE_FATAL;
		}
		memcpy(a->mem + *a->mem_used, buf, len);
		*a->mem_used += len;
		return ARCHIVE_OK;
	}
	const uint8_t *p = buf;
	while (len) {
		ssize_t w = write(a->fd, p, len);
		if (w <= 0) {
			archive_set_error(&a->archive, errno, "Write error");
			return ARCHIVE_FATAL;
		}
		p += w; len -= (size_t)w;
	}
	return ARCHIVE_OK;
}

struct archive_write_filter *__archive_write_allocate_filter(struct archive *_a)
{
	struct archive_write *a = (struct archive_write *)_a;
	struct archive_write_filter *f = calloc(1, sizeof(*f));
	if (!f)
		return NULL;
	f->archive = _a;
	f->state = 1;	/* ARCHIVE_WRITE_FILTER_STATE_NEW */
	if (a->filter_first == NULL)
		a->filter_first = f;
	else
		a->filter_last->next_filter = f;
	a->filter_last = f;
	return f;
}

int __archive_write_filter(struct archive_write_filter *f, const void *buf, size_t len)
{
	if (len == 0)
		return ARCHIVE_OK;
	if (f == NULL || f->write == NULL)
		return ARCHIVE_FATAL;
	int r = f->write(f, buf, len);
	f->bytes_written += (int64_t)len;
	return r;
}

struct archive *archive_write_new(void)
{
	struct archive_write *a = calloc(1, sizeof(*a));
	if (a) {
		a->archive.state = LA_STATE_NEW;
		a->fd = -1;
	}
	return (struct archive *)a;
}

int archive_write_set_format_raw(struct archive *_a)
{
	((struct archive_write *)_a)->format_raw = 1;
	_a->archive_format = ARCHIVE_FORMAT_RAW;
	_a->archive_format_name = "raw";
	return ARCHIVE_OK;
}

int archive_write_set_filter_option(struct archive *_a, const char *m, const char *o, const char *v)
{
	struct archive_write *a = (struct archive_write *)_a;
	int handled = 0;
	for (struct archive_write_filter *f = a->filter_first; f; f = f->next_filter) {
		if (f->options == NULL || (m != NULL && (f->name == NULL || strcmp(f->name, m) != 0)))
			continue;
		int r = f->options(f, o, v);
		if (r == ARCHIVE_FATAL || r == ARCHIVE_FAILED)
			return r;
		if (r == ARCHIVE_OK)
			handled = 1;
	}
	if (!handled) {
		archive_set_error(_a, ARCHIVE_ERRNO_MISC, "Undefined option: `%s%s%s'", m ? m : "", m ? ":" : "", o);
		return ARCHIVE_FAILED;
	}
	return ARCHIVE_OK;
}

static int write_open_common(struct archive_write *a)
{
	/* the client sink is the last "filter" of the chain */
	struct archive_write_filter *sink = __archive_write_allocate_filter(&a->archive);
	if (!sink) {
		archive_set_error(&a->archive, ENOMEM, "Out of memory");
		return ARCHIVE_FATAL;
	}
	sink->write = client_write;
	sink->name = "client";
	for (struct archive_write_filter *f = a->filter_first; f; f = f->next_filter) {
		if (f->open) {
			int r = f->open(f);
			if (r != ARCHIVE_OK)
				return r;
		}
		f->state = 2;	/* OPEN */
	}
	a->opened = 1;
	a->archive.state = LA_STATE_HEADER;
	return ARCHIVE_OK;
}

int archive_write_open_memory(struct archive *_a, void *buff, size_t size, size_t *used)
{
	struct archive_write *a = (struct archive_write *)_a;
	a->mem = buff; a->mem_cap = size; a->mem_used = used;
	*used = 0;
	return write_open_common(a);
}

int archive_write_open_fd(struct archive *_a, int fd)
{
	struct archive_write *a = (struct archive_write *)_a;
	a->fd = fd;
	return write_open_common(a);
}

int archive_write_header(struct archive *_a, struct archive_entry *entry)
{
	struct archive_write *a = (struct archive_write *)_a;
	(void)entry;
	if (!a->opened || !a->format_raw) {
		archive_set_error(_a, ARCHIVE_ERRNO_MISC, "No format defined (this slice writes the raw format)");
		return ARCHIVE_FATAL;
	}
	if (a->entries++ > 0) {
		archive_set_error(_a, ERANGE, "Raw format only supports one entry per archive");	/* archive_write_set_format_raw.c:80-84 */
		return ARCHIVE_FATAL;
	}
	_a->state = LA_STATE_DATA;
	return ARCHIVE_OK;
}

ssize_t archive_write_data(struct archive *_a, const void *buff, size_t s)
{
	struct archive_write *a = (struct archive_write *)_a;
	if (!a->opened || a->entries == 0) {
		archive_set_error(_a, ARCHIVE_ERRNO_MISC, "archive_write_data before archive_write_header");
		return ARCHIVE_FATAL;
	}
	int r = __archive_write_filter(a->filter_first, buff, s);
	return r == ARCHIVE_OK ? (ssize_t)s : r;
}

int archive_write_close(struct archive *_a)
{
	struct archive_write *a = (struct archive_write *)_a;
	int rc = ARCHIVE_OK;
	if (a->closed || !a->opened)
		return ARCHIVE_OK;
	for (struct archive_write_filter *f = a->filter_first; f; f = f->next_filter) {
		if (f->close) {
			int r = f->close(f);
			if (r < rc)
				rc = r;
		}
		f->state = 4;	/* CLOSED */
	}
	a->closed = 1;
	_a->state = LA_STATE_CLOSED;
	return rc;
}

int archive_write_free(struct archive *_a)
{
	struct archive_write *a = (struct archive_write *)_a;
	if (!a)
		return ARCHIVE_OK;
	int rc = archive_write_close(_a);
	struct archive_write_filter *f = a->filter_first;
	while (f) {
		struct archive_write_filter *n = f->next_filter;
		if (f->free)
			f->free(f);
		free(f);
		f = n;
	}
	free(a);
	return rc;
}

/* ------------------------------------------------------------------ the lz4 write filter */

#define LZ4W_BLOCK 65536u
#define LZ4W_BPF   16u
#define GZW_CHUNK  49152u

struct lz4w_private {	/* archive_write_add_filter_lz4.c:49-68; the gzip filter shares the window machinery */
	int kind;		/* 0 = lz4, 1 = gzip */
	int timestamp;		/* gzip: > 0 writes time(NULL) into the member headers (archive_write_add_filter_gzip.c:213-220) */
	uint32_t mtime;
	int compression_level;
	unsigned block_independence:1, block_checksum:1, stream_checksum:1;
	unsigned block_maximum_size:3;
	la_gpu_ctx *gpu;
	uint8_t *win;		/* pinned input window */
	size_t win_cap, win_len;
	uint8_t *out;		/* pinned output of one window */
	size_t out_cap;
	void *d_in, *d_out, *d_len;
	size_t d_in_cap, d_out_cap;
	int wrote_anything;
	int64_t total_in;
};

static int lz4w_options(struct archive_write_filter *f, const char *key, const char *value)
{
	struct lz4w_private *d = f->data;
	if (strcmp(key, "compression-level") == 0) {
		if (value == NULL || !(value[0] >= '1' && value[0] <= '9') || value[1] != '\0')
			return ARCHIVE_WARN;
		d->compression_level = value[0] - '0';	/* (the device has one level) */
		return ARCHIVE_OK;
	}
	if (strcmp(key, "stream-checksum") == 0) { d->stream_checksum = value != NULL; return ARCHIVE_OK; }
	if (strcmp(key, "block-checksum") == 0) { d->block_checksum = value != NULL; return ARCHIVE_OK; }
	if (strcmp(key, "block-size") == 0) {
		if (value == NULL || !(value[0] >= '4' && value[0] <= '7') || value[1] != '\0')
			return ARCHIVE_WARN;
		d->block_maximum_size = (unsigned)(value[0] - '0');	/* (accepted; blocks are 64 KiB) */
		return ARCHIVE_OK;
	}
	if (strcmp(key, "block-dependence") == 0) {
		if (value != NULL) {
			archive_set_error(f->archive, ARCHIVE_ERRNO_MISC, "block dependence is not supported by the GPU lz4 writer");
			return ARCHIVE_FAILED;
		}
		return ARCHIVE_OK;
	}
	return ARCHIVE_WARN;
}

static int lz4w_gpu_fail(struct archive_write_filter *f, struct lz4w_private *d, const char *what)
{
	archive_set_error(f->archive, ARCHIVE_ERRNO_MISC, "%s GPU data plane: %s failed: %s", d->kind ? "gzip" : "lz4", what,
	    d->gpu ? la_gpu_last_error(d->gpu) : "no device");
	return ARCHIVE_FATAL;
}

/* compress the window and hand the frames to the next filter */
static int lz4w_flush_window(struct archive_write_filter *f, struct lz4w_private *d)
{
	if (d->win_len == 0)
		return ARCHIVE_OK;
	const uint32_t flags = (d->block_checksum ? LA_LZ4C_BLOCK_SUM : 0) | (d->stream_checksum ? LA_LZ4C_CONTENT_SUM : 0);
	const uint64_t bound = d->kind ? la_gpu_gzip_compress_bound(d->win_len, GZW_CHUNK) : la_gpu_lz4_compress_bound(d->win_len, LZ4W_BLOCK, LZ4W_BPF);
	if (d->d_in_cap < d->win_len) {
		if (d->d_in) la_gpu_free(d->gpu, d->d_in);
		d->d_in = NULL; d->d_in_cap = 0;
		if (la_gpu_malloc(d->gpu, &d->d_in, d->win_cap) != LA_OK)
			return lz4w_gpu_fail(f, d, "device allocation");
		d->d_in_cap = d->win_cap;
	}
	if (d->d_out_cap < bound) {
		if (d->d_out) la_gpu_free(d->gpu, d->d_out);
		if (d->out) la_gpu_free_host(d->gpu, d->out);
		d->d_out = NULL; d->out = NULL; d->d_out_cap = d->out_cap = 0;
		const uint64_t cap = d->kind ? la_gpu_gzip_compress_bound(d->win_cap, GZW_CHUNK) : la_gpu_lz4_compress_bound(d->win_cap, LZ4W_BLOCK, LZ4W_BPF);
		void *hp = NULL;
		if (la_gpu_malloc(d->gpu, &d->d_out, cap) != LA_OK || la_gpu_malloc_host(d->gpu, &hp, cap) != LA_OK)
			return lz4w_gpu_fail(f, d, "output allocation");
		d->out = hp; d->d_out_cap = d->out_cap = cap;
	}
	if (d->d_len == NULL && la_gpu_malloc(d->gpu, &d->d_len, 64) != LA_OK)
		return lz4w_gpu_fail(f, d, "device allocation");
	la_lz4c_batch bt;
	memset(&bt, 0, sizeof(bt));
	bt.d_src = d->d_in; bt.src_bytes = d->win_len;
	bt.block_size = LZ4W_BLOCK; bt.blocks_per_frame = LZ4W_BPF; bt.flags = flags;
	bt.d_out = d->d_out; bt.out_cap = d->d_out_cap; bt.d_out_bytes = d->d_len;
	la_gzc_batch gt;
	memset(&gt, 0, sizeof(gt));
	gt.d_src = d->d_in; gt.src_bytes = d->win_len; gt.chunk_bytes = GZW_CHUNK; gt.mtime = d->mtime;
	gt.d_out = d->d_out; gt.out_cap = d->d_out_cap; gt.d_out_bytes = d->d_len;
	uint64_t total = 0;
	if (la_gpu_memcpy_h2d(d->gpu, d->d_in, d->win, d->win_len) != LA_OK ||
	    (d->kind ? la_gpu_gzip_compress(d->gpu, &gt) : la_gpu_lz4_compress(d->gpu, &bt)) != LA_OK ||
	    la_gpu_memcpy_d2h(d->gpu, &total, d->d_len, sizeof(total)) != LA_OK ||
	    la_gpu_sync(d->gpu) != LA_OK)
		return lz4w_gpu_fail(f, d, "compress");
	if (total > d->out_cap)
		return lz4w_gpu_fail(f, d, "compress (output bound)");
	if (la_gpu_memcpy_d2h(d->gpu, d->out, d->d_out, total) != LA_OK || la_gpu_sync(d->gpu) != LA_OK)
		return lz4w_gpu_fail(f, d, "device to host copy");
	d->win_len = 0;
	d->wrote_anything = 1;
	return __archive_write_filter(f->next_filter, d->out, (size_t)total);
}

static int lz4w_write(struct archive_write_filter *f, const void *buff, size_t length)
{
	struct lz4w_private *d = f->data;
	const uint8_t *p = buff;
	d->total_in += (int64_t)length;
	while (length) {
		size_t n = d->win_cap - d->win_len;
		if (n > length) n = length;
		memcpy(d->win + d->win_len, p, n);
		d->win_len += n; p += n; length -= n;
		if (d->win_len == d->win_cap) {
			int r = lz4w_flush_window(f, d);
			if (r != ARCHIVE_OK)
				return r;
		}
	}
	return ARCHIVE_OK;
}

static int lz4w_open(struct archive_write_filter *f)
{
	struct lz4w_private *d = f->data;
	const char *dev = getenv("LA_GPU_DEVICE"), *wm = getenv("LA_GPU_WRITE_WINDOW_MIB");
	if (la_gpu_open(dev ? atoi(dev) : 0, &d->gpu) != LA_OK) {
		archive_set_error(f->archive, ARCHIVE_ERRNO_MISC,
		    "Can't initialize %s GPU data plane (no usable gfx950 device); no CPU fallback is built", d->kind ? "gzip" : "lz4");
		return ARCHIVE_FATAL;
	}
	d->win_cap = (size_t)(wm && atoi(wm) > 0 ? atoi(wm) : 64) << 20;	/* (a multiple of the 1 MiB frame) */
	void *hp = NULL;
	if (la_gpu_malloc_host(d->gpu, &hp, d->win_cap) != LA_OK)
		return lz4w_gpu_fail(f, d, "pinned window allocation");
	d->win = hp;
	if (d->kind && d->timestamp >= 0)
		d->mtime = (uint32_t)time(NULL);
	f->write = lz4w_write;
	return ARCHIVE_OK;
}

static int lz4w_close(struct archive_write_filter *f)
{
	struct lz4w_private *d = f->data;
	if (d->gpu == NULL)
		return ARCHIVE_OK;
	int r = lz4w_flush_window(f, d);
	if (r == ARCHIVE_OK && !d->wrote_anything && d->kind) {
		/* nothing was written: one member with an empty deflate stream, as zlib's Z_FINISH on no input gives
		 * the reference (header, 03 00, crc 0, isize 0) */
		const uint8_t m[20] = { 0x1f, 0x8b, 8, 0, (uint8_t)d->mtime, (uint8_t)(d->mtime >> 8), (uint8_t)(d->mtime >> 16),
		    (uint8_t)(d->mtime >> 24), 0, 3, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0 };
		r = __archive_write_filter(f->next_filter, m, sizeof(m));
	} else if (r == ARCHIVE_OK && !d->wrote_anything) {
		/* nothing was written: one empty frame (header, EndMark, checksum of nothing), as the
		 * reference's close does (archive_write_add_filter_lz4.c:300-330) */
		uint8_t h[15];
		const uint8_t flg = (uint8_t)(0x60 | (d->block_checksum ? 0x10 : 0) | (d->stream_checksum ? 0x04 : 0));
		h[0] = 0x04; h[1] = 0x22; h[2] = 0x4D; h[3] = 0x18; h[4] = flg; h[5] = 0x40;
		h[6] = (uint8_t)(la_archive_xxhash.XXH32(h + 4, 2, 0) >> 8);
		memset(h + 7, 0, 4);
		size_t n = 11;
		if (d->stream_checksum) {
			const uint32_t c = la_archive_xxhash.XXH32("", 0, 0);
			h[11] = (uint8_t)c; h[12] = (uint8_t)(c >> 8); h[13] = (uint8_t)(c >> 16); h[14] = (uint8_t)(c >> 24);
			n = 15;
		}
		r = __archive_write_filter(f->next_filter, h, n);
	}
	return r;
}

static int lz4w_free(struct archive_write_filter *f)
{
	struct lz4w_private *d = f->data;
	if (d) {
		if (d->gpu) {
			la_gpu_sync(d->gpu);
			if (d->win) la_gpu_free_host(d->gpu, d->win);
			if (d->out) la_gpu_free_host(d->gpu, d->out);
			if (d->d_in) la_gpu_free(d->gpu, d->d_in);
			if (d->d_out) la_gpu_free(d->gpu, d->d_out);
			if (d->d_len) la_gpu_free(d->gpu, d->d_len);
			la_gpu_close(d->gpu);
		}
		free(d);
	}
	f->data = NULL;
	return ARCHIVE_OK;
}

int archive_write_add_filter_lz4(struct archive *_a)
{
	struct archive_write_filter *f = __archive_write_allocate_filter(_a);
	struct lz4w_private *d = calloc(1, sizeof(*d));
	if (f == NULL || d == NULL) {
		free(d);
		archive_set_error(_a, ENOMEM, "Out of memory");
		return ARCHIVE_FATAL;
	}
	d->compression_level = 1;
	d->block_independence = 1;
	d->block_checksum = 0;
	d->stream_checksum = 1;
	d->block_maximum_size = 7;
	f->data = d;
	f->options = lz4w_options;
	f->open = lz4w_open;
	f->close = lz4w_close;
	f->free = lz4w_free;
	f->code = ARCHIVE_FILTER_LZ4;
	f->name = "lz4";
	return ARCHIVE_OK;
}

/* ------------------------------------------------------------------ the gzip write filter */

static int gzw_options(struct archive_write_filter *f, const char *key, const char *value)
{
	struct lz4w_private *d = f->data;
	if (strcmp(key, "compression-level") == 0) {	/* archive_write_add_filter_gzip.c:147-153 */
		if (value == NULL || !(value[0] >= '0' && value[0] <= '9') || value[1] != '\0')
			return ARCHIVE_WARN;
		d->compression_level = value[0] - '0';	/* (the device has one level) */
		return ARCHIVE_OK;
	}
	if (strcmp(key, "timestamp") == 0) {		/* :154-157 */
		d->timestamp = (value == NULL) ? -1 : 1;
		return ARCHIVE_OK;
	}
	return ARCHIVE_WARN;
}

int archive_write_add_filter_gzip(struct archive *_a)
{
	struct archive_write_filter *f = __archive_write_allocate_filter(_a);
	struct lz4w_private *d = calloc(1, sizeof(*d));
	if (f == NULL || d == NULL) {
		free(d);
		archive_set_error(_a, ENOMEM, "Out of memory");
		return ARCHIVE_FATAL;
	}
	d->kind = 1;
	d->compression_level = 6;	/* Z_DEFAULT_COMPRESSION in the reference; informational here */
	f->data = d;
	f->options = gzw_options;
	f->open = lz4w_open;
	f->close = lz4w_close;
	f->free = lz4w_free;
	f->code = ARCHIVE_FILTER_GZIP;
	f->name = "gzip";
	return ARCHIVE_OK;
}
