/*
 * la_zstd_index.c -- host-side walk of a Zstandard stream: frame and block headers only (RFC 8878 3.1.1), no
 * entropy decoding.  It cuts the image into whole frames for la_gpu_zstd_decode (include/la_gpu.h), drops skippable
 * frames (the reference's bidder accepts them, archive_read_support_filter_zstd.c:117-130, and libzstd skips them)
 * and gives every frame an output slot: Frame_Content_Size when the header carries one, otherwise the sum over its
 * blocks of Block_Size (raw, RLE) or 128 KiB (compressed: Block_Maximum_Size).
 */
#include "la_host.h"
#include <string.h>

static uint32_t le24(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
static uint32_t le32(const uint8_t *p) { return le24(p) | ((uint32_t)p[3] << 24); }

int la_zstd_bid_bytes(const uint8_t *p, size_t avail)
{
	/* zstd.c:107-131 */
	if (avail < 4)
		return 0;
	const uint32_t m = le32(p);
	if (m == 0xFD2FB528u || (m & 0xFFFFFFF0u) == 0x184D2A50u)
		return 32;
	return 0;
}

/* One frame at p[0..len).  Returns 1 whole frame (*flen, *bound set; *bound = 0 and *skippable = 1 for a skippable
 * frame), 0 the frame is not complete inside len, -1 not a frame (unknown magic). */
static int zstd_frame_extent(const uint8_t *p, uint64_t len, uint64_t *flen, uint64_t *bound, int *skippable, int *bad_block)
{
	*skippable = 0;
	*bad_block = 0;
	if (len < 4)
		return 0;
	const uint32_t magic = le32(p);
	if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
		if (len < 8)
			return 0;
		const uint64_t sz = 8ull + le32(p + 4);
		if (sz > len)
			return 0;
		*flen = sz;
		*bound = 0;
		*skippable = 1;
		return 1;
	}
	if (magic != 0xFD2FB528u)
		return -1;
	if (len < 5)
		return 0;
	const int fhd = p[4], fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, csum = (fhd >> 2) & 1, did_flag = fhd & 3;
	static const int did_len[4] = { 0, 1, 2, 4 };
	const int fcs_len = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
	uint64_t q = 5 + (single ? 0 : 1) + (uint64_t)did_len[did_flag];
	if (q + (uint64_t)fcs_len > len)
		return 0;
	uint64_t fcs = 0;
	for (int i = 0; i < fcs_len; i++)
		fcs |= (uint64_t)p[q + i] << (8 * i);
	if (fcs_len == 2)
		fcs += 256;
	q += (uint64_t)fcs_len;
	uint64_t sum = 0;
	for (;;) {
		if (q + 3 > len)
			return 0;
		const uint32_t bh = le24(p + q);
		q += 3;
		const int last = bh & 1, type = (bh >> 1) & 3;
		const uint32_t bsize = bh >> 3;
		if (type == 3 || bsize > 128u * 1024u) {	/* the device names the error; nothing behind it can be found */
			*bad_block = 1;
			*flen = q;
			*bound = (fcs_len && fcs < sum) ? fcs : sum;	/* (never the bare claim: it may be forged) */
			return 1;
		}
		const uint64_t body = type == 1 ? 1 : bsize;
		if (q + body > len)
			return 0;
		q += body;
		sum += type == 2 ? 128u * 1024u : bsize;
		if (last)
			break;
	}
	if (csum) {
		if (q + 4 > len)
			return 0;
		q += 4;
	}
	*flen = q;
	/* a claimed content size is checked by the device against what the blocks produce; never trust it beyond the
	 * blocks' own bound (a forged size must not reserve memory) */
	*bound = (fcs_len && fcs < sum) ? fcs : sum;
	return 1;
}

int la_zstd_index_build(const uint8_t *img, uint64_t len, int at_eof, uint64_t out_budget, la_zstd_frame *frames,
    uint32_t cap, la_zstd_index_result *res)
{
	uint64_t p = 0, out = 0;
	uint32_t n = 0;
	memset(res, 0, sizeof(*res));
	res->end_kind = LA_END_EOF;
	while (p < len) {
		uint64_t flen = 0, bound = 0;
		int skippable = 0, bad_block = 0;
		const int r = zstd_frame_extent(img + p, len - p, &flen, &bound, &skippable, &bad_block);
		if (r == 0) {
			res->end_kind = at_eof ? LA_END_TRUNCATED : LA_END_NEED_MORE;
			break;
		}
		if (r < 0) {
			res->end_kind = LA_END_ZSTD_BAD_MAGIC;
			break;
		}
		if (!skippable) {
			if (n == cap || (out_budget && n && out + bound > out_budget)) {
				res->end_kind = LA_END_NEED_MORE;	/* the rest is the next window's */
				res->window_full = 1;
				break;
			}
			frames[n].src_off = p;
			frames[n].src_len = flen;
			frames[n].dst_off = out;
			frames[n].dst_cap = bound;
			out += (bound + 15u) & ~(uint64_t)15u;
			n++;
		}
		p += flen;
		if (bad_block) {
			res->end_kind = LA_END_ZSTD_BAD_BLOCK;
			break;
		}
	}
	if (p == len && !at_eof && res->end_kind == LA_END_EOF)
		res->end_kind = LA_END_NEED_MORE;	/* a frame boundary, but the input may go on */
	res->n_frames = n;
	res->consumed = p;
	res->dst_bytes = out;
	return 0;
}
