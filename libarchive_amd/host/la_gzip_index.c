/*
 * la_gzip_index.c -- host walker for .gz images: member table for the device.
 *
 * Restates the FRAMING of libarchive/archive_read_support_filter_gzip.c only:
 * header parse :128-239 (magic 1f 8b 08, reserved flag bits, FEXTRA / FNAME /
 * FCOMMENT / FHCRC skipping, mtime + name capture) and the member loop
 * :446-501 (members repeat; anything that is not a header ends the stream
 * silently).  gzip members do not say how long they are, so the
 * walker finds the NEXT member either
 *   - exactly, from a BGZF-style "BC" extra subfield (total size - 1, u16), or
 *   - speculatively, at the next 1f 8b 08 with valid flag bits; the device
 *     decode then confirms or refutes the boundary (the deflate stream must
 *     end exactly 8 bytes before it).
 * No deflate bit is looked at here.
 */
#define _GNU_SOURCE
#include "../../include/la_host.h"
#include <stdlib.h>
#include <string.h>

static uint32_t le32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

size_t la_gz_header_parse(const uint8_t *p, size_t avail, la_gz_header *h)
{
	size_t len = 10;
	if (h) memset(h, 0, sizeof(*h));
	if (avail < 10)
		return 0;
	if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 0x08)
		return 0;
	if (p[3] & 0xE0)
		return 0;
	int flags = p[3];
	if (h) h->mtime = le32(p + 4);
	if (flags & 4) {
		if (avail < len + 2)
			return 0;
		size_t xlen = ((size_t)p[len + 1] << 8) | p[len];
		/* BGZF "BC" subfield: SI1 SI2 SLEN(2) BSIZE(2) */
		if (h && avail >= len + 2 + xlen) {
			size_t q = len + 2, e = len + 2 + xlen;
			while (q + 4 <= e) {
				size_t sl = ((size_t)p[q + 3] << 8) | p[q + 2];
				if (p[q] == 'B' && p[q + 1] == 'C' && sl == 2 && q + 6 <= e)
					h->bgzf_size = (((uint32_t)p[q + 5] << 8) | p[q + 4]) + 1u;
				q += 4 + sl;
			}
		}
		len += xlen + 2;
	}
	if (flags & 8) {
		size_t start = len;
		do {
			++len;
			if (avail < len)
				return 0;
		} while (p[len - 1] != 0);
		if (h) h->name_off = (uint32_t)start;
	}
	if (flags & 16) {
		do {
			++len;
			if (avail < len)
				return 0;
		} while (p[len - 1] != 0);
	}
	if (flags & 2) {
		if (avail < len + 2)
			return 0;
		len += 2;
	}
	if (h) h->len = (uint32_t)len;
	return len;
}

int la_gz_bid_bytes(const uint8_t *p, size_t avail)
{
	return la_gz_header_parse(p, avail, NULL) ? 27 : 0;
}

void la_gz_index_free(la_gz_index *x)
{
	if (!x) return;
	free(x->members); free(x->headers);
	memset(x, 0, sizeof(*x));
}

static int push(la_gz_index *x)
{
	if (x->n == x->cap) {
		uint32_t nc = x->cap ? x->cap * 2 : 256;
		la_gz_member *m = realloc(x->members, (size_t)nc * sizeof(*m));
		if (!m) return -1;
		x->members = m;
		la_gz_header *h = realloc(x->headers, (size_t)nc * sizeof(*h));
		if (!h) return -1;
		x->headers = h;
		x->cap = nc;
	}
	x->n++;
	return 0;
}

/* first offset >= from with the bytes 1f 8b 08, or len: the one pass over the window that finds plain gzip
 * members (they do not say how long they are).  Two compares per 32 bytes where AVX2 is there, memchr otherwise. */
#if defined(__x86_64__)
#include <immintrin.h>

/* Largest compressed span of ONE member the tables can express (32-bit fields).  LA_GZ_TEST_SPAN_LIMIT lowers it
 * so that the refusal path (LA_END_GZ_TOO_LARGE, "gzip member too large ...") can be reached by a test without a
 * 4 GiB input; it cannot raise it. */
uint64_t la_gz_span_limit(void)
{
	const char *v = getenv("LA_GZ_TEST_SPAN_LIMIT");
	uint64_t lim = 0xFFFFFFFFull;
	if (v != NULL && strtoull(v, NULL, 10) > 0 && strtoull(v, NULL, 10) < lim)
		lim = strtoull(v, NULL, 10);
	return lim;
}
__attribute__((target("avx2")))
static uint64_t find_magic_avx2(const uint8_t *p, uint64_t len, uint64_t i)
{
	const __m256i a = _mm256_set1_epi8(0x1f), b = _mm256_set1_epi8((char)0x8b);
	for (; i + 34 <= len; i += 32) {
		__m256i x = _mm256_loadu_si256((const __m256i *)(p + i));
		__m256i y = _mm256_loadu_si256((const __m256i *)(p + i + 1));
		unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpeq_epi8(x, a), _mm256_cmpeq_epi8(y, b)));
		while (m) {
			unsigned k = (unsigned)__builtin_ctz(m);
			m &= m - 1;
			if (p[i + k + 2] == 0x08)
				return i + k;
		}
	}
	for (; i + 3 <= len; i++)
		if (p[i] == 0x1f && p[i + 1] == 0x8b && p[i + 2] == 0x08)
			return i;
	return len;
}
#endif

static uint64_t find_magic(const uint8_t *p, uint64_t len, uint64_t from)
{
#if defined(__x86_64__)
	static int have_avx2 = -1;
	if (have_avx2 < 0)
		have_avx2 = (__builtin_cpu_supports("avx2") && getenv("LA_NO_AVX2") == NULL) ? 1 : 0;
	if (have_avx2)
		return find_magic_avx2(p, len, from);
#endif
	while (from + 3 <= len) {
		const uint8_t *hit = memchr(p + from, 0x1f, (size_t)(len - from - 2));
		if (!hit)
			return len;
		uint64_t o = (uint64_t)(hit - p);
		if (p[o + 1] == 0x8b && p[o + 2] == 0x08)
			return o;
		from = o + 1;
	}
	return len;
}

/* next offset >= from where a plausible member header starts, or len */
static uint64_t next_candidate(const uint8_t *img, uint64_t len, uint64_t from, int strict)
{
	while (from + 4 <= len) {
		uint64_t o = find_magic(img, len, from);
		if (o >= len)
			return len;
		if (o + 4 <= len && (img[o + 3] & 0xE0) == 0) {
			/* XFL (byte 8) and OS (byte 9) as gzip, zlib, pigz, bgzip, Java and Go write them; a header cut
			 * by the window edge passes (the walker then asks for more input) */
			if (!strict || o + 10 > len ||
			    ((img[o + 8] == 0 || img[o + 8] == 2 || img[o + 8] == 4) && (img[o + 9] <= 13 || img[o + 9] == 255)))
				return o;
		}
		from = o + 1;
	}
	return len;
}

#define LA_GZ_MAX_SLOT 0x80000000u

int la_gz_index_build2(const uint8_t *img, uint64_t len, int at_eof, uint32_t first_skip,
    uint32_t first_cap, la_gz_index *x)
{
	return la_gz_index_build3(img, len, at_eof, first_skip, first_cap, 0, x);
}

int la_gz_index_build3(const uint8_t *img, uint64_t len, int at_eof, uint32_t first_skip,
    uint32_t first_cap, uint32_t flags, la_gz_index *x)
{
	return la_gz_index_build4(img, len, at_eof, first_skip, first_cap, flags, 0, x);
}

/* out_budget != 0 bounds the window by DECODED bytes too (sum of the members' output slots): a window of
 * highly compressible members would otherwise claim up to 1032 x its size in slab.  The walker stops in front
 * of the member that would pass the budget (at least one member is always taken) and reports
 * LA_END_NEED_MORE: the rest of the window is the next window's. */
int la_gz_index_build4(const uint8_t *img, uint64_t len, int at_eof, uint32_t first_skip,
    uint32_t first_cap, uint32_t flags, uint64_t out_budget, la_gz_index *x)
{
	const int strict = (flags & LA_GZ_INDEX_STRICT) != 0;
	uint64_t pos = 0, out = 0;
	memset(x, 0, sizeof(*x));
	x->end_kind = LA_END_EOF;

	for (;;) {
		x->consumed = pos;
		if (out_budget && x->n > 0 && out >= out_budget) {
			x->end_kind = LA_END_NEED_MORE;
			break;
		}
		la_gz_header h;
		size_t hlen = la_gz_header_parse(img + pos, (size_t)(len - pos), &h);
		if (hlen == 0) {
			/* Not a header: the stream ends silently here (gzip.c:351-353) -- unless this
			 * is a window and the bytes could be a header cut short by its edge. */
			uint64_t rem = len - pos;
			int maybe = rem < 3 ? (rem == 0 || (img[pos] == 0x1f && (rem < 2 || img[pos + 1] == 0x8b)))
			    : (img[pos] == 0x1f && img[pos + 1] == 0x8b && img[pos + 2] == 0x08 &&
			       (rem < 4 || (img[pos + 3] & 0xE0) == 0));
			x->end_kind = (!at_eof && maybe) ? LA_END_NEED_MORE : LA_END_EOF;
			break;
		}
		h.off = pos;
		uint64_t body = pos + hlen;
		if (body >= len) {
			/* header runs to the end: at EOF the reference reports truncated input */
			x->end_kind = at_eof ? LA_END_TRUNCATED : LA_END_NEED_MORE;
			if (at_eof) {
				if (push(x) < 0) return -1;
				x->headers[x->n - 1] = h;
				la_gz_member *m = &x->members[x->n - 1];
				m->src_off = len; m->src_len = 0; m->dst_cap = 0; m->dst_off = out;
				x->consumed = len;
			}
			break;
		}
		uint64_t next;
		if (h.bgzf_size && h.bgzf_size >= hlen + 8 && !(x->n == 0 && first_skip)) {
			next = pos + h.bgzf_size;
			if (next > len) {
				if (!at_eof) { x->end_kind = LA_END_NEED_MORE; break; }
				next = len;	/* truncated member: the decode reports it */
			}
		} else {
			next = next_candidate(img, len, body + 2, strict);
			/* boundaries the decode already refuted for this member are skipped */
			for (uint32_t k = 0; x->n == 0 && k < first_skip && next < len; k++)
				next = next_candidate(img, len, next + 1, strict);
			x->speculative = 1;
			if (next == len && !at_eof) {
				/* cannot tell whether this member is complete: wait for more input
				 * (a lone member larger than the window makes the caller widen it) */
				x->end_kind = LA_END_NEED_MORE;
				break;
			}
		}
		uint64_t span = next - body;
		if (span > la_gz_span_limit()) {
			x->end_kind = LA_END_GZ_TOO_LARGE;	/* > 4 GiB member: beyond the table's u32 fields */
			break;
		}
		if (push(x) < 0) return -1;
		x->headers[x->n - 1] = h;
		la_gz_member *m = &x->members[x->n - 1];
		m->src_off = body;
		m->src_len = (uint32_t)span;
		uint32_t isize = span >= 8 ? le32(img + next - 4) : 0;
		/* slot from the ISIZE claim; a deflate stream cannot expand more than ~1032x */
		uint64_t bound = span * 1032 + 64;
		if (isize > bound) isize = (uint32_t)(bound > 0xFFFFFFFFull ? 0xFFFFFFFFull : bound);
		if (isize > LA_GZ_MAX_SLOT) isize = LA_GZ_MAX_SLOT;	/* (32-bit positions in the kernels: op + 258 must not wrap) */
		if (x->n == 1 && first_cap > isize)
			isize = first_cap;	/* the decode found the ISIZE claim too small */
		m->dst_cap = isize;
		m->dst_off = out;
		out += isize;
		pos = next;
		if (pos >= len) {
			x->consumed = pos;
			x->end_kind = at_eof ? LA_END_EOF : LA_END_NEED_MORE;
			break;
		}
	}
	x->max_out = out;
	return 0;
}

int la_gz_index_build(const uint8_t *img, uint64_t len, int at_eof, la_gz_index *x)
{
	return la_gz_index_build2(img, len, at_eof, 0, 0, x);
}
