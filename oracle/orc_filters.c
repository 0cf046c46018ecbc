/*
 * orc_filters.c -- ORACLE (test infrastructure, see la_oracle.h): what the
 * reference's lz4 and gzip read filters deliver for a whole file image.
 *
 * Restates, call by call, the state machines of
 *   libarchive/archive_read_support_filter_lz4.c:289-721  and
 *   libarchive/archive_read_support_filter_gzip.c:128-239, :340-511
 * over an in-memory image (the reference sees the same bytes through
 * __archive_read_filter_ahead/consume).  One oracle "read call" = one call of the
 * filter's vtable read(); a call that returns 0 ends the stream for the read
 * core (archive_read.c:1394-1411), a negative call is fatal and its partial
 * output is lost.
 */
#include "la_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define ARCHIVE_FATAL (-30)

static uint32_t le32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* ------------------------------------------------------------------ lz4 */

#define LZ4_MAGIC    0x184D2204u
#define LZ4_SKIP     0x184D2A50u
#define LZ4_LEGACY   0x184C2102u
#define LEGACY_BLOCK (8 * 1024 * 1024)
#define LEGACY_BOUND (LEGACY_BLOCK + LEGACY_BLOCK / 255 + 16)

int orc_lz4_bid(const uint8_t *p, size_t avail)
{
	if (avail < 11)	/* lz4.c:150 */
		return 0;
	uint32_t m = le32(p);
	if (m == LZ4_MAGIC) {
		if (((p[4] & 0xc0) >> 6) != 1) return 0;
		if (p[4] & 2) return 0;
		if (((p[5] & 0x70) >> 4) < 4) return 0;
		if (p[5] & ~0x70) return 0;
		return 48;
	}
	if (m == LZ4_LEGACY)
		return 32;
	return 0;
}

enum { ST_SELECT, ST_DEF_STREAM, ST_DEF_BLOCK, ST_LEG_STREAM, ST_LEG_BLOCK };

typedef struct {
	const uint8_t *src; size_t len, pos;
	int stage, eof;
	int indep, bsum, ssum, bmax;
	size_t unconsumed;
	size_t decoded_size;	/* lz4.c:78, drives the prefix slide */
	uint8_t *blk;		/* [64 KiB prefix | block] */
	size_t blk_size;
	orc_xxh32_state xs;
	orc_stream_result *res;
} lz4st;

/* bench.py's cpu_baseline leg can put the box's own liblz4 (dlopen'ed there) behind the same
 * framing: that IS what libarchive's filter executes (lz4.c:557-561, :578-584).  NULL = the port. */
typedef int (*ext_lz4_safe_fn)(const char *, char *, int, int);
typedef int (*ext_lz4_dict_fn)(const char *, char *, int, int, const char *, int);
static ext_lz4_safe_fn ext_lz4_safe;
static ext_lz4_dict_fn ext_lz4_dict;
void orc_set_external_lz4(void *safe, void *using_dict)
{
	ext_lz4_safe = (ext_lz4_safe_fn)safe;
	ext_lz4_dict = (ext_lz4_dict_fn)using_dict;
}
static int lz4_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap, const uint8_t *dict, int dict_len)
{
	if (!dict && ext_lz4_safe)
		return ext_lz4_safe((const char *)src, (char *)dst, src_len, dst_cap);
	if (dict && ext_lz4_dict)
		return ext_lz4_dict((const char *)src, (char *)dst, src_len, dst_cap, (const char *)dict, dict_len);
	return orc_lz4_block_decode(src, src_len, dst, dst_cap, dict, dict_len);
}

static int lz4_fail(lz4st *s, const char *msg)
{
	snprintf(s->res->errmsg, sizeof(s->res->errmsg), "%s", msg);
	return ARCHIVE_FATAL;
}
static size_t lz4_left(const lz4st *s) { return s->len - s->pos; }

static int lz4_grow(lz4st *s, size_t need)
{
	if (s->blk_size < need) {
		free(s->blk);
		s->blk = malloc(need);
		s->blk_size = need;
		if (!s->blk) return -1;
	}
	return 0;
}

/* lz4.c:370-469 */
static int lz4_descriptor(lz4st *s)
{
	if (lz4_left(s) < 2)
		return lz4_fail(s, "truncated lz4 input");
	const uint8_t *p = s->src + s->pos;
	uint8_t flag = p[0], bd = p[1];
	if ((flag & 0xc0) != 0x40 || (flag & 0x02))
		return lz4_fail(s, "malformed lz4 data");
	s->indep = (flag & 0x20) != 0;
	s->bsum = (flag & 0x10) ? 4 : 0;
	s->ssum = (flag & 0x04) != 0;
	if (bd & 0x8f)
		return lz4_fail(s, "malformed lz4 data");
	switch (bd >> 4) {
	case 4: s->bmax = 64 * 1024; break;
	case 5: s->bmax = 256 * 1024; break;
	case 6: s->bmax = 1024 * 1024; break;
	case 7: s->bmax = 4 * 1024 * 1024; break;
	default: return lz4_fail(s, "malformed lz4 data");
	}
	size_t dbytes = 3 + ((flag & 0x08) ? 8 : 0) + ((flag & 0x01) ? 4 : 0);
	if (lz4_left(s) < dbytes)
		return lz4_fail(s, "truncated lz4 input");
	if (((orc_xxh32(p, dbytes - 1, 0) >> 8) & 0xff) != p[dbytes - 1])
		return lz4_fail(s, "malformed lz4 data");
	s->pos += dbytes;
	/* lz4.c:240-263 */
	size_t need = (size_t)s->bmax + (s->indep ? 0 : 65536);
	if (lz4_grow(s, need + 65536) < 0)
		return lz4_fail(s, "Can't allocate data for lz4 decompression");
	if (!s->indep)
		memset(s->blk, 0, 65536);
	if (s->ssum)
		orc_xxh32_init(&s->xs, 0);
	s->decoded_size = 0;
	return 0;
}

/* lz4.c:471-613.  Returns bytes (>=0) with *p set, or ARCHIVE_FATAL. */
static long lz4_data_block(lz4st *s, const uint8_t **p)
{
	*p = NULL;
	if (lz4_left(s) < 4)
		return lz4_fail(s, "truncated lz4 input");
	const uint8_t *rb = s->src + s->pos;
	uint32_t w = le32(rb);
	if ((w & 0x7fffffffu) > (uint32_t)s->bmax)
		return lz4_fail(s, "malformed lz4 data");
	if (w == 0) {
		s->pos += 4;
		return 0;
	}
	size_t csize = w & 0x7fffffffu;
	size_t usize = (w & 0x80000000u) ? csize : 0;
	if (lz4_left(s) < 4 + csize + (size_t)s->bsum)
		return lz4_fail(s, "truncated lz4 input");
	if (s->bsum && orc_xxh32(rb + 4, csize, 0) != le32(rb + 4 + csize))
		return lz4_fail(s, "malformed lz4 data");

	if (usize) {
		if (!s->indep) {
			if (usize < 65536) {
				memcpy(s->blk + 65536 - usize, rb + 4, usize);
				memset(s->blk, 0, 65536 - usize);
			} else
				memcpy(s->blk, rb + 4 + usize - 65536, 65536);
			s->decoded_size = 0;
		}
		s->unconsumed = 4 + usize + (size_t)s->bsum;
		*p = rb + 4;
		return (long)usize;
	}

	int n;
	size_t prefix = 0;
	if (s->indep) {
		n = lz4_block(rb + 4, (int)csize, s->blk, s->bmax, NULL, 0);
	} else {
		prefix = 65536;
		if (s->decoded_size) {
			if (s->decoded_size < prefix) {
				memmove(s->blk + prefix - s->decoded_size, s->blk + prefix, s->decoded_size);
				memset(s->blk, 0, prefix - s->decoded_size);
			} else
				memmove(s->blk, s->blk + s->decoded_size, prefix);
		}
		n = lz4_block(rb + 4, (int)csize, s->blk + prefix, s->bmax, s->blk, (int)prefix);
	}
	if (n < 0)
		return lz4_fail(s, "lz4 decompression failed");
	s->unconsumed = 4 + csize + (size_t)s->bsum;
	*p = s->blk + prefix;
	s->decoded_size = (size_t)n;
	s->res->n_units++;
	return n;
}

/* lz4.c:615-668 */
static long lz4_default_stream(lz4st *s, const uint8_t **p)
{
	long ret;
	if (s->stage == ST_SELECT) {
		s->stage = ST_DEF_STREAM;
		if ((ret = lz4_descriptor(s)) != 0)
			return ret;
		s->stage = ST_DEF_BLOCK;
		s->res->n_frames++;
	}
	ret = lz4_data_block(s, p);
	if (ret == 0 && *p == NULL)
		s->stage = ST_SELECT;
	if (s->ssum) {
		if (s->stage == ST_SELECT) {
			if (lz4_left(s) < 4)
				return lz4_fail(s, "truncated lz4 input");
			uint32_t want = le32(s->src + s->pos);
			s->pos += 4;
			if (want != orc_xxh32_digest(&s->xs))
				return lz4_fail(s, "lz4 stream checksum error");
		} else if (ret > 0)
			orc_xxh32_update(&s->xs, *p, (size_t)ret);
	}
	return ret;
}

/* lz4.c:670-721 */
static long lz4_legacy_stream(lz4st *s, const uint8_t **p)
{
	*p = NULL;
	if (lz4_grow(s, LEGACY_BLOCK) < 0)
		return lz4_fail(s, "Can't allocate data for lz4 decompression");
	if (lz4_left(s) < 4) {
		if (s->stage == ST_SELECT) {
			s->stage = ST_LEG_STREAM;
			return lz4_fail(s, "truncated lz4 input");
		}
		s->stage = ST_SELECT;
		return 0;
	}
	if (s->stage == ST_SELECT)
		s->res->n_frames++;
	s->stage = ST_LEG_BLOCK;
	uint32_t csize = le32(s->src + s->pos);
	if (csize > LEGACY_BOUND) {
		s->stage = ST_SELECT;
		return 0;
	}
	if (lz4_left(s) < 4 + (size_t)csize)
		return lz4_fail(s, "truncated lz4 input");
	int n = lz4_block(s->src + s->pos + 4, (int)csize, s->blk,
	    (int)(s->blk_size > 0x7fffffff ? 0x7fffffff : s->blk_size), NULL, 0);
	if (n < 0)
		return lz4_fail(s, "lz4 decompression failed");
	*p = s->blk;
	s->unconsumed = 4 + (size_t)csize;
	s->res->n_units++;
	return n;
}

/* lz4.c:289-368 */
static long lz4_read(lz4st *s, const uint8_t **p)
{
	long ret;
	if (s->eof) { *p = NULL; return 0; }
	s->pos += s->unconsumed;
	s->unconsumed = 0;

	switch (s->stage) {
	case ST_SELECT:
		break;
	case ST_DEF_STREAM:
	case ST_LEG_STREAM:
		return lz4_fail(s, "Invalid sequence.");
	case ST_DEF_BLOCK:
		ret = lz4_default_stream(s, p);
		if (ret != 0 || s->stage != ST_SELECT)
			return ret;
		break;
	case ST_LEG_BLOCK:
		ret = lz4_legacy_stream(s, p);
		if (ret != 0 || s->stage != ST_SELECT)
			return ret;
		break;
	}
	while (s->stage == ST_SELECT) {
		if (lz4_left(s) < 4) {
			s->eof = 1; *p = NULL; return 0;
		}
		uint32_t m = le32(s->src + s->pos);
		s->pos += 4;
		if (m == LZ4_MAGIC)
			return lz4_default_stream(s, p);
		else if (m == LZ4_LEGACY)
			return lz4_legacy_stream(s, p);
		else if ((m & ~0xFu) == LZ4_SKIP) {
			if (lz4_left(s) < 4)
				return lz4_fail(s, "Malformed lz4 data");
			uint64_t skip = 4 + (uint64_t)le32(s->src + s->pos);
			/* a consume past the end is not checked by the filter (lz4.c:356);
			 * the next ahead() then reports end of input */
			if (skip > lz4_left(s))
				s->pos = s->len;
			else
				s->pos += (size_t)skip;
		} else {
			s->eof = 1; *p = NULL; return 0;
		}
	}
	s->eof = 1; *p = NULL;
	return 0;
}

int orc_lz4_stream_decode(const uint8_t *src, size_t src_len,
    uint8_t *out, size_t out_cap, orc_stream_result *res)
{
	lz4st s;
	memset(&s, 0, sizeof(s));
	memset(res, 0, sizeof(*res));
	s.src = src; s.len = src_len; s.res = res;
	s.stage = ST_SELECT;
	for (;;) {
		const uint8_t *p = NULL;
		long n = lz4_read(&s, &p);
		if (n < 0) { res->rc = ARCHIVE_FATAL; break; }
		if (n == 0) break;
		if (out) {
			if (res->out_len + (size_t)n > out_cap) { free(s.blk); return -1; }
			memcpy(out + res->out_len, p, (size_t)n);
		}
		res->out_len += (size_t)n;
	}
	free(s.blk);
	return 0;
}

/* ------------------------------------------------------------------ gzip */

/* gzip.c:128-239 over a flat image.  Returns header length or 0. */
size_t orc_gzip_header_len(const uint8_t *p, size_t avail, uint32_t *mtime,
    char *name, size_t name_cap, int *has_name)
{
	size_t len = 10;
	if (avail < 10)
		return 0;
	if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 0x08)
		return 0;
	if (p[3] & 0xE0)
		return 0;
	int flags = p[3];
	if (mtime) *mtime = le32(p + 4);
	if (flags & 4) {
		if (avail < len + 2) return 0;
		len += ((size_t)p[len + 1] << 8) | p[len];
		len += 2;
	}
	if (flags & 8) {
		size_t start = len;
		do {
			++len;
			if (avail < len) return 0;
		} while (p[len - 1] != 0);
		if (name && name_cap) {
			snprintf(name, name_cap, "%s", (const char *)p + start);
			if (has_name) *has_name = 1;
		}
	}
	if (flags & 16) {
		do {
			++len;
			if (avail < len) return 0;
		} while (p[len - 1] != 0);
	}
	if (flags & 2) {
		if (avail < len + 2) return 0;
		len += 2;
	}
	return len;
}

int orc_gzip_bid(const uint8_t *p, size_t avail)
{
	return orc_gzip_header_len(p, avail, NULL, NULL, 0, NULL) ? 27 : 0;
}

int orc_gzip_stream_decode(const uint8_t *src, size_t src_len,
    uint8_t *out, size_t out_cap, orc_stream_result *res)
{
	size_t pos = 0;
	size_t total = 0;		/* bytes produced so far (delivered or pending) */
	size_t delivered;
	int first_call_done = 0;	/* metadata snapshot taken */
	uint32_t mtime = 0; char name[256]; int has_name = 0;
	const size_t CH = 65536;
	size_t scratch_cap = 0; uint8_t *scratch = NULL;

	memset(res, 0, sizeof(*res));
	name[0] = 0;

	for (;;) {
		/* gzip.c:449-457: header of the next member, or silent end */
		uint32_t mt = 0; char nm[256]; int hn = 0;
		size_t hlen = orc_gzip_header_len(src + pos, src_len - pos, &mt, nm, sizeof(nm), &hn);
		if (hlen == 0)
			break;		/* ARCHIVE_EOF: trailing garbage / end of file */
		/* this header is parsed during the FIRST read() iff the output
		 * block was not yet full when the loop came round (gzip.c:446) */
		if (total < CH) {
			mtime = mt;	/* peek_at_header overwrites mtime each time, :162 */
			if (hn) { memcpy(name, nm, sizeof(name)); has_name = 1; }
		}
		if (hlen >= src_len - pos) {
			/* header runs to/past the end: consume fails silently, ahead(1) is NULL */
			snprintf(res->errmsg, sizeof(res->errmsg), "truncated gzip input");
			res->rc = ARCHIVE_FATAL;
			delivered = (total / CH) * CH;
			goto done;
		}
		pos += hlen;

		size_t consumed = 0, produced = 0;
		const uint8_t *body = src + pos;
		size_t body_len = src_len - pos;
		uint8_t *dst;
		size_t cap;
		if (out) {
			dst = out + total; cap = out_cap - total;
		} else {
			/* count-only mode: decode into a growing scratch buffer */
			size_t want = body_len * 1040 + 65536;
			if (want > ((size_t)1 << 31)) want = (size_t)1 << 31;
			if (scratch_cap < want) { free(scratch); scratch = malloc(want); scratch_cap = want; }
			dst = scratch; cap = scratch_cap;
		}
		int rc = orc_inflate_raw(body, body_len, dst, cap, &consumed, &produced);
		if (rc == ORC_INF_OUT_FULL) { free(scratch); return -1; }
		res->n_units++;
		if (rc == ORC_INF_TRUNCATED) {
			total += produced;
			snprintf(res->errmsg, sizeof(res->errmsg), "truncated gzip input");
			res->rc = ARCHIVE_FATAL;
			delivered = (total / CH) * CH;
			goto done;
		}
		if (rc == ORC_INF_DATA_ERROR) {
			/* zlib runs ahead of a full output block until it needs to store a
			 * byte, so the error is raised by the read() that emitted the
			 * last good byte of this member (gzip error order) */
			snprintf(res->errmsg, sizeof(res->errmsg), "gzip decompression failed");
			res->rc = ARCHIVE_FATAL;
			if (produced == 0)
				delivered = (total / CH) * CH;
			else
				delivered = ((total + produced - 1) / CH) * CH;
			total += produced;
			goto done;
		}
		pos += consumed;
		/* gzip.c:398-429: 8-byte trailer, never verified by the reference */
		if (src_len - pos < 8) {
			res->errmsg[0] = 0;	/* ARCHIVE_FATAL with no message (F11 v) */
			res->rc = ARCHIVE_FATAL;
			if (produced == 0)
				delivered = (total / CH) * CH;
			else
				delivered = ((total + produced - 1) / CH) * CH;
			total += produced;
			goto done;
		}
		if (le32(src + pos) != orc_crc32(0, dst, produced) ||
		    le32(src + pos + 4) != (uint32_t)produced)
			res->gz_trailer_mismatch = 1;
		pos += 8;
		total += produced;
		(void)first_call_done;
	}
	delivered = total;
done:
	res->out_len = delivered;
	res->gz_mtime = mtime;
	res->gz_has_name = has_name;
	memcpy(res->gz_name, name, sizeof(res->gz_name));
	free(scratch);
	return 0;
}
