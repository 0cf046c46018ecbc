/*
 * orc_inflate.c -- ORACLE (test infrastructure, see la_oracle.h): raw DEFLATE.
 *
 * The arithmetic is NOT in the reference tree: libarchive's gzip filter calls
 * zlib's inflate() after inflateInit2(-15) (archive_read_support_filter_gzip.c
 * :363, :479).  This restates RFC 1951 (stored / fixed / dynamic blocks,
 * LSB-first bit packing, canonical Huffman codes of at most 15 bits, 32 KiB
 * distance limit) with zlib 1.2.11's accept/reject rules:
 *   - block type 3, LEN != ~NLEN, HLIT > 286, HDIST > 30 are errors;
 *   - the code-length code must be complete; literal/length and distance codes
 *     may be incomplete only when their longest code is 1 bit; over-subscribed
 *     sets are errors; a missing end-of-block code (length 0) is an error;
 *   - repeat code 16 with no previous length, or a repeat running past
 *     HLIT+HDIST, is an error;
 *   - literal/length symbols 286/287, distance symbols 30/31, an unassigned
 *     code, or a distance reaching before the start of THIS stream's output
 *     are errors.
 * On error or truncation *produced is the number of bytes zlib would have
 * emitted before noticing (whole symbols only; stored data byte-wise).
 */
#include "la_oracle.h"
#include <string.h>

typedef struct {
	const uint8_t *src;
	size_t len, pos;
	uint64_t hold;
	int bits;
} bitrd;

/* returns 0 on success, -1 if the input is exhausted */
static int need(bitrd *b, int n)
{
	while (b->bits < n) {
		if (b->pos >= b->len)
			return -1;
		b->hold |= (uint64_t)b->src[b->pos++] << b->bits;
		b->bits += 8;
	}
	return 0;
}
static uint32_t take(bitrd *b, int n)
{
	uint32_t v = (uint32_t)(b->hold & ((1ull << n) - 1));
	b->hold >>= n;
	b->bits -= n;
	return v;
}

typedef struct {
	uint16_t count[16];	/* codes per length */
	uint16_t symbol[288];	/* symbols ordered by code */
	int max_len;
} hufftab;

/* returns 0 complete, >0 incomplete (unused code space), <0 over-subscribed */
static int huff_build(hufftab *h, const uint8_t *lens, int n)
{
	uint16_t offs[16];
	int left = 1;

	memset(h->count, 0, sizeof(h->count));
	for (int i = 0; i < n; i++)
		h->count[lens[i]]++;
	h->max_len = 0;
	for (int l = 15; l >= 1; l--)
		if (h->count[l]) { h->max_len = l; break; }
	for (int l = 1; l <= 15; l++) {
		left <<= 1;
		left -= h->count[l];
		if (left < 0)
			return -1;
	}
	offs[1] = 0;
	for (int l = 1; l < 15; l++)
		offs[l + 1] = offs[l] + h->count[l];
	for (int i = 0; i < n; i++)
		if (lens[i])
			h->symbol[offs[lens[i]]++] = (uint16_t)i;
	return left;
}

/* decode one symbol: >=0 symbol, -1 input exhausted, -2 unassigned code.
 * An empty table behaves like zlib's one-bit "invalid code" entry. */
static int huff_decode(bitrd *b, const hufftab *h)
{
	int code = 0, first = 0, index = 0;
	int ml = h->max_len ? h->max_len : 1;
	for (int l = 1; l <= ml; l++) {
		if (need(b, 1) < 0)
			return -1;
		code |= (int)take(b, 1);
		int cnt = h->count[l];
		if (code - cnt < first)
			return h->symbol[index + (code - first)];
		index += cnt;
		first += cnt;
		first <<= 1;
		code <<= 1;
	}
	return -2;
}

static const uint16_t len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
	35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
	3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
	257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
	7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
static const uint8_t clc_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

static int inflate_codes(bitrd *b, uint8_t *dst, size_t cap, size_t *op,
    const hufftab *lt, const hufftab *dt)
{
	for (;;) {
		int sym = huff_decode(b, lt);
		if (sym == -1) return ORC_INF_TRUNCATED;
		if (sym < 0) return ORC_INF_DATA_ERROR;
		if (sym < 256) {
			if (*op >= cap) return ORC_INF_OUT_FULL;
			dst[(*op)++] = (uint8_t)sym;
			continue;
		}
		if (sym == 256)
			return ORC_INF_OK;
		sym -= 257;
		if (sym >= 29) return ORC_INF_DATA_ERROR;
		if (need(b, len_extra[sym]) < 0) return ORC_INF_TRUNCATED;
		size_t length = len_base[sym] + take(b, len_extra[sym]);
		int ds = huff_decode(b, dt);
		if (ds == -1) return ORC_INF_TRUNCATED;
		if (ds < 0 || ds >= 30) return ORC_INF_DATA_ERROR;
		if (need(b, dist_extra[ds]) < 0) return ORC_INF_TRUNCATED;
		size_t dist = dist_base[ds] + take(b, dist_extra[ds]);
		if (dist > *op) return ORC_INF_DATA_ERROR;
		if (*op + length > cap) return ORC_INF_OUT_FULL;
		for (size_t i = 0; i < length; i++, (*op)++)
			dst[*op] = dst[*op - dist];
	}
}

int orc_inflate_raw(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
    size_t *consumed, size_t *produced)
{
	bitrd b = { src, src_len, 0, 0, 0 };
	size_t op = 0;
	int rc = ORC_INF_OK;
	static hufftab fixed_l, fixed_d;
	static int fixed_ready;

	if (!fixed_ready) {
		uint8_t l[288];
		int i = 0;
		for (; i < 144; i++) l[i] = 8;
		for (; i < 256; i++) l[i] = 9;
		for (; i < 280; i++) l[i] = 7;
		for (; i < 288; i++) l[i] = 8;
		huff_build(&fixed_l, l, 288);
		for (i = 0; i < 32; i++) l[i] = 5;
		huff_build(&fixed_d, l, 32);	/* symbols 30/31 decode, then fail the >= 30 check */
		fixed_ready = 1;
	}

	for (;;) {
		if (need(&b, 3) < 0) { rc = ORC_INF_TRUNCATED; break; }
		int last = (int)take(&b, 1);
		int type = (int)take(&b, 2);

		if (type == 0) {
			take(&b, b.bits & 7);	/* to byte boundary */
			if (need(&b, 32) < 0) { rc = ORC_INF_TRUNCATED; break; }
			uint32_t v = take(&b, 32);
			uint32_t len = v & 0xffff, nlen = v >> 16;
			if (len != (nlen ^ 0xffff)) { rc = ORC_INF_DATA_ERROR; break; }
			/* hand whole bytes still in the bit buffer back, then copy
			 * straight from the byte stream */
			b.pos -= (size_t)(b.bits >> 3);
			b.bits = 0; b.hold = 0;
			size_t avail = b.len - b.pos;
			size_t n = len < avail ? len : avail;
			if (op + n > dst_cap) { rc = ORC_INF_OUT_FULL; break; }
			memcpy(dst + op, b.src + b.pos, n);
			op += n; b.pos += n;
			if (n < len) { rc = ORC_INF_TRUNCATED; break; }
		} else if (type == 1) {
			rc = inflate_codes(&b, dst, dst_cap, &op, &fixed_l, &fixed_d);
			if (rc != ORC_INF_OK) break;
		} else if (type == 2) {
			uint8_t lens[320];
			hufftab cl, lt, dt;
			if (need(&b, 14) < 0) { rc = ORC_INF_TRUNCATED; break; }
			int nlen = (int)take(&b, 5) + 257;
			int ndist = (int)take(&b, 5) + 1;
			int ncode = (int)take(&b, 4) + 4;
			if (nlen > 286 || ndist > 30) { rc = ORC_INF_DATA_ERROR; break; }
			memset(lens, 0, 19);
			int i;
			for (i = 0; i < ncode; i++) {
				if (need(&b, 3) < 0) break;
				lens[clc_order[i]] = (uint8_t)take(&b, 3);
			}
			if (i < ncode) { rc = ORC_INF_TRUNCATED; break; }
			if (huff_build(&cl, lens, 19) != 0 && cl.max_len != 0) { rc = ORC_INF_DATA_ERROR; break; }
			int idx = 0;
			rc = ORC_INF_OK;
			if (cl.max_len == 0) {
				/* zlib 1.2.11 builds a one-bit "invalid" table for an all-zero
				 * code-length code and the length reader takes its value 0:
				 * one bit per length, then the missing end-of-block is caught. */
				if (need(&b, 1) < 0) { rc = ORC_INF_TRUNCATED; break; }
				for (; idx < nlen + ndist; idx++) {
					if (need(&b, 1) < 0) break;
					take(&b, 1);
					lens[idx] = 0;
				}
				if (idx < nlen + ndist) { rc = ORC_INF_TRUNCATED; break; }
			}
			while (idx < nlen + ndist) {
				int sym = huff_decode(&b, &cl);
				if (sym == -1) { rc = ORC_INF_TRUNCATED; break; }
				if (sym < 0) { rc = ORC_INF_DATA_ERROR; break; }
				if (sym < 16) {
					lens[idx++] = (uint8_t)sym;
					continue;
				}
				int rep, val = 0;
				if (sym == 16) {
					if (need(&b, 2) < 0) { rc = ORC_INF_TRUNCATED; break; }
					if (idx == 0) { rc = ORC_INF_DATA_ERROR; break; }
					val = lens[idx - 1];
					rep = 3 + (int)take(&b, 2);
				} else if (sym == 17) {
					if (need(&b, 3) < 0) { rc = ORC_INF_TRUNCATED; break; }
					rep = 3 + (int)take(&b, 3);
				} else {
					if (need(&b, 7) < 0) { rc = ORC_INF_TRUNCATED; break; }
					rep = 11 + (int)take(&b, 7);
				}
				if (idx + rep > nlen + ndist) { rc = ORC_INF_DATA_ERROR; break; }
				while (rep--)
					lens[idx++] = (uint8_t)val;
			}
			if (rc != ORC_INF_OK) break;
			if (lens[256] == 0) { rc = ORC_INF_DATA_ERROR; break; }
			int e = huff_build(&lt, lens, nlen);
			if (e < 0 || (e > 0 && lt.max_len != 1)) { rc = ORC_INF_DATA_ERROR; break; }
			e = huff_build(&dt, lens + nlen, ndist);
			if (e < 0 || (e > 0 && dt.max_len > 1)) { rc = ORC_INF_DATA_ERROR; break; }
			rc = inflate_codes(&b, dst, dst_cap, &op, &lt, &dt);
			if (rc != ORC_INF_OK) break;
		} else {
			rc = ORC_INF_DATA_ERROR;
			break;
		}
		if (last)
			break;
	}
	if (consumed) {
		/* whole unused bytes still in the bit buffer go back to the input */
		*consumed = b.pos - (size_t)(b.bits >> 3);
	}
	if (produced)
		*produced = op;
	return rc;
}
