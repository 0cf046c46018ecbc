/*
 * la_bid_policy.c -- should the GPU filter take this stream at all?
 *
 * The device decodes INDEPENDENT units in parallel (lz4 frames / blocks, gzip members, zstd frames); ONE serial
 * unit -- a plain single-member .gz (gzip.c:431-511), a one-frame .zst (zstd.c:196-260), a single lz4 frame whose
 * content checksum is one XXH32 chain (lz4.c:615-668) -- runs on one wave and is 1.5 to 100 times slower than the
 * reference's own filter on one host core (known limits).  The product has no CPU path, so the honest
 * answer for such a stream is NOT TO BID: with libarchive's own bidder registered beside this one
 * (archive_read.c:557-565 takes the highest bid) the reference's filter then decodes it.
 *
 * The decision is made in bid() from a bounded look-ahead (bid() may only peek, archive_read_private.h:43-51):
 *   - fewer bytes than the look-ahead in the whole stream: small, taken (nothing to win or lose);
 *   - evidence of many units inside the look-ahead (a BGZF size subfield, a second member header, a frame that
 *     ends inside it): taken;
 *   - otherwise the first unit is larger than the look-ahead: declined (bid 0).
 * LA_GPU_BID=all switches the policy off (every stream with the right magic is taken: the stand-alone test core
 * has no other bidder); LA_GPU_BID_LOOKAHEAD_KIB sets the look-ahead (default 1024, gzip 256).
 */
#include "la_read_private.h"
#include "la_host.h"
#include <stdlib.h>
#include <string.h>

int la_bid_take_all(void)
{
	const char *m = getenv("LA_GPU_BID");
	return m != NULL && strcmp(m, "all") == 0;
}

size_t la_bid_lookahead(size_t dflt_kib)
{
	const char *v = getenv("LA_GPU_BID_LOOKAHEAD_KIB");
	size_t kib = dflt_kib;
	if (v != NULL && atol(v) > 0)
		kib = (size_t)atol(v);
	if (kib > (64u << 10))
		kib = 64u << 10;
	return kib << 10;
}

/* as many bytes as upstream has, at most `want`; *got = how many (0: none) */
const unsigned char *la_bid_peek(struct archive_read_filter *filter, size_t want, size_t *got)
{
	ssize_t avail = 0;
	const unsigned char *p = __archive_read_filter_ahead(filter, want, &avail);
	if (p == NULL && avail > 0)
		p = __archive_read_filter_ahead(filter, (size_t)avail, &avail);
	if (p == NULL || avail <= 0) {
		*got = 0;
		return NULL;
	}
	*got = (size_t)avail < want ? (size_t)avail : want;
	return p;
}

/* gzip: p[0..n) starts with a member whose header (hdr_len bytes, parsed by the caller) carries no BGZF size.
 * 1 = take it.  A second member header inside the look-ahead counts as evidence of a many-member stream; the
 * candidate test is the strict one of the boundary search (XFL / OS bytes that real writers emit). */
int la_bid_gzip_parallel(const unsigned char *p, size_t n, size_t hdr_len, size_t lookahead)
{
	if (n < lookahead)
		return 1;	/* the whole stream is smaller than the look-ahead */
	for (size_t i = hdr_len + 1; i + 10 <= n; i++) {
		const unsigned char *q = memchr(p + i, 0x1f, n - 10 - i + 1);
		if (q == NULL)
			break;
		i = (size_t)(q - p);
		if (q[1] == 0x8b && q[2] == 0x08 && (q[3] & 0xE0) == 0 && (q[8] == 0 || q[8] == 2 || q[8] == 4) &&
		    (q[9] <= 13 || q[9] == 255))
			return 1;
	}
	return 0;
}

/* lz4: 1 = take it.  A frame without a content checksum decodes block-parallel whatever its size; a frame WITH one is
 * bound by its XXH32 chain, so it has to end inside the look-ahead (then the stream is made of frames that small). */
int la_bid_lz4_parallel(const unsigned char *p, size_t n, size_t lookahead)
{
	if (n < lookahead || n < 11)
		return 1;
	const uint32_t magic = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
	if (magic != 0x184D2204u)
		return 1;	/* legacy frame: no checksums */
	const unsigned flg = p[4];
	if (!(flg & 0x04))
		return 1;	/* no content checksum */
	size_t pos = 4 + 3 + ((flg & 0x08) ? 8 : 0) + ((flg & 0x01) ? 4 : 0);
	const int bsum = (flg & 0x10) != 0;
	while (pos + 4 <= n) {
		const uint32_t w = (uint32_t)p[pos] | (uint32_t)p[pos + 1] << 8 | (uint32_t)p[pos + 2] << 16 | (uint32_t)p[pos + 3] << 24;
		if (w == 0)
			return 1;	/* EndMark inside the look-ahead */
		pos += 4 + (size_t)(w & 0x7FFFFFFFu) + (bsum ? 4 : 0);
	}
	return 0;
}

/* zstd: 1 = take it: the first frame (skippable frames in front of it passed over) ends inside the look-ahead */
int la_bid_zstd_parallel(const unsigned char *p, size_t n, size_t lookahead)
{
	if (n < lookahead)
		return 1;
	la_zstd_frame fr[4];
	la_zstd_index_result r;
	memset(&r, 0, sizeof(r));
	if (la_zstd_index_build(p, n, 0, ~0ull, fr, 4, &r) != 0)
		return 1;	/* let the filter report what is wrong with it */
	return r.n_frames >= 1;
}
