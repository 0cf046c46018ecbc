/*
 * la_lz4_index.c -- host walker for .lz4 images: builds the block / frame job
 * tables the device decodes (include/la_gpu.h).
 *
 * Restates the FRAMING of libarchive/archive_read_support_filter_lz4.c only:
 * frame selection :328-364, descriptor parse :370-469 (the check byte itself is
 * verified on the device), block size words :485-514, EndMark / content
 * checksum position :493-496 + :639-651, legacy frames :670-721.  No payload
 * byte is hashed or decoded here.
 */
#include "../../include/la_host.h"
#include <stdlib.h>
#include <string.h>

#define LZ4_MAGIC    0x184D2204u
#define LZ4_SKIP     0x184D2A50u
#define LZ4_LEGACY   0x184C2102u
#define LEGACY_BLOCK (8u * 1024 * 1024)
#define LEGACY_BOUND (LEGACY_BLOCK + LEGACY_BLOCK / 255 + 16)

static uint32_t le32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

int la_lz4_bid_bytes(const uint8_t *p, size_t avail)
{
	if (avail < 11)
		return 0;
	uint32_t m = le32(p);
	if (m == LZ4_MAGIC) {
		if (((p[4] & 0xc0) >> 6) != 1 || (p[4] & 2))
			return 0;
		if (((p[5] & 0x70) >> 4) < 4 || (p[5] & ~0x70))
			return 0;
		return 48;
	}
	return m == LZ4_LEGACY ? 32 : 0;
}

static la_lz4_block *push_block(la_lz4_index *x)
{
	if (x->n_blocks == x->cap_blocks) {
		uint32_t nc = x->cap_blocks ? x->cap_blocks * 2 : 1024;
		la_lz4_block *nb = realloc(x->blocks, (size_t)nc * sizeof(*nb));
		if (!nb) return NULL;
		x->blocks = nb; x->cap_blocks = nc;
	}
	return &x->blocks[x->n_blocks++];
}
static la_lz4_frame *push_frame(la_lz4_index *x)
{
	if (x->n_frames == x->cap_frames) {
		uint32_t nc = x->cap_frames ? x->cap_frames * 2 : 64;
		la_lz4_frame *nf = realloc(x->frames, (size_t)nc * sizeof(*nf));
		if (!nf) return NULL;
		x->frames = nf; x->cap_frames = nc;
	}
	la_lz4_frame *f = &x->frames[x->n_frames++];
	memset(f, 0, sizeof(*f));
	return f;
}

void la_lz4_index_free(la_lz4_index *x)
{
	if (!x) return;
	free(x->blocks); free(x->frames);
	memset(x, 0, sizeof(*x));
}

/*
 * A window that ends inside an item is "need more" unless at_eof, in which
 * case it is what the reference reports when ahead() comes back NULL.
 */
#define SHORT(kind_at_eof)                                             \
	do {                                                           \
		x->end_kind = at_eof ? (kind_at_eof) : LA_END_NEED_MORE; \
		goto out;                                              \
	} while (0)

int la_lz4_index_build(const uint8_t *img, uint64_t len, int at_eof, la_lz4_index *x)
{
	return la_lz4_index_build2(img, len, at_eof, NULL, x);
}

int la_lz4_index_build2(const uint8_t *img, uint64_t len, int at_eof, la_lz4_resume *rs, la_lz4_index *x)
{
	return la_lz4_index_build3(img, len, at_eof, rs, 0, x);
}

/* out_budget != 0 bounds the window by DECODED bytes too (sum of the blocks' dst_cap): highly
 * compressible input would otherwise make one window of compressed bytes ask for tens of GiB of
 * slab (the reference streams with one block buffer, lz4.c:240-263).  The walker stops in front of
 * the block that would pass the budget -- inside a frame only with a resume record -- and reports
 * LA_END_NEED_MORE; at least one block is always taken. */
int la_lz4_index_build3(const uint8_t *img, uint64_t len, int at_eof, la_lz4_resume *rs, uint64_t out_budget,
    la_lz4_index *x)
{
	uint64_t pos = 0;
	memset(x, 0, sizeof(*x));
	x->end_kind = LA_END_EOF;
	int resume = rs && rs->in_frame == 1;	/* the window starts inside a frame */
	int resume_legacy = rs && rs->in_frame == 2;	/* ... inside a legacy frame */

	for (;;) {
		/* lz4.c:328-364: select the next stream */
		x->consumed = pos;
		if (out_budget && x->n_blocks > 0 && x->max_out >= out_budget && !resume && !resume_legacy) {
			x->end_kind = LA_END_NEED_MORE;	/* the window is full by decoded bytes: the next frame waits */
			goto out;
		}
		uint32_t m;
		if (resume)
			m = LZ4_MAGIC;
		else if (resume_legacy)
			m = LZ4_LEGACY;
		else {
			if (len - pos < 4)
				SHORT(LA_END_EOF);
			m = le32(img + pos);
		}
		if (m == LZ4_MAGIC) {
			uint64_t p = pos + 4;
			uint32_t bmax, dbytes = 0, bsum;
			int indep, ssum, cont = resume;
			uint32_t blocks_before = 0;
			resume = 0;
			if (cont) {
				/* continuation of a frame that began in an earlier window */
				p = pos;
				bmax = rs->bmax;
				indep = !(rs->flags & 4u);
				bsum = (rs->flags & 1u) ? 4 : 0;
				ssum = (rs->flags & 2u) != 0;
				blocks_before = rs->blocks_so_far;
				rs->in_frame = 0;
			} else {
			if (len - p < 2)
				SHORT(LA_END_TRUNCATED);
			uint8_t flag = img[p], bd = img[p + 1];
			if ((flag & 0xc0) != 0x40 || (flag & 0x02) || (bd & 0x8f)) {
				x->end_kind = LA_END_MALFORMED; goto out;
			}
			switch (bd >> 4) {
			case 4: bmax = 64u << 10; break;
			case 5: bmax = 256u << 10; break;
			case 6: bmax = 1u << 20; break;
			case 7: bmax = 4u << 20; break;
			default: x->end_kind = LA_END_MALFORMED; goto out;
			}
			dbytes = 3 + ((flag & 0x08) ? 8 : 0) + ((flag & 0x01) ? 4 : 0);
			if (len - p < dbytes)
				SHORT(LA_END_TRUNCATED);
			indep = (flag & 0x20) != 0;
			bsum = (flag & 0x10) ? 4 : 0;
			ssum = (flag & 0x04) != 0;
			}

			/* tentatively index the frame; roll back if the window ends inside it */
			uint32_t save_b = x->n_blocks, save_f = x->n_frames;
			uint64_t save_out = x->max_out;
			la_lz4_frame *f = push_frame(x);
			if (!f) return -1;
			uint32_t fi = x->n_frames - 1;
			f->desc_off = cont ? 0 : p;
			f->desc_len = dbytes;
			f->first_block = x->n_blocks;
			f->flags = (cont ? LA_LZ4F_CONT : LA_LZ4F_HEADER_SUM) | (ssum ? (LA_LZ4F_CONTENT_SUM | LA_LZ4F_HASHED) : 0);
			p += dbytes;
			int end = -1, budget_stop = 0;
			for (;;) {
				if (len - p < 4) { end = LA_END_TRUNCATED; break; }
				uint32_t w = le32(img + p);
				if ((w & 0x7fffffffu) > bmax) { x->end_kind = LA_END_MALFORMED; x->consumed = p; goto frame_cut; }
				if (w == 0) {
					p += 4;
					if (ssum) {
						if (len - p < 4) {
							end = LA_END_TRUNCATED;
							if (!at_eof)
								p -= 4;	/* the EndMark is read again with the next window */
							break;
						}
						x->frames[fi].content_sum = le32(img + p);
						p += 4;
					}
					break;
				}
				uint32_t csize = w & 0x7fffffffu;
				if (len - p < 4ull + csize + bsum) { end = LA_END_TRUNCATED; break; }
				if (out_budget && rs && x->frames[fi].n_blocks > 0 && x->max_out + bmax > out_budget) {
					end = LA_END_TRUNCATED; budget_stop = 1; break;	/* this block opens the next window */
				}
				la_lz4_block *b = push_block(x);
				if (!b) return -1;
				b->src_off = p + 4;
				b->src_len = csize;
				b->dst_cap = bmax;
				b->flags = ((w & 0x80000000u) && csize ? LA_LZ4B_STORED : 0) |
				    (bsum ? LA_LZ4B_CHECKSUM : 0) | (indep ? 0 : LA_LZ4B_DEPENDENT) |
				    (x->n_blocks - 1 == x->frames[fi].first_block ? (cont ? (indep ? 0 : LA_LZ4B_HIST) : LA_LZ4B_FIRST) : 0);
				b->block_sum = bsum ? le32(img + p + 4 + csize) : 0;
				x->max_out += bmax;
				p += 4ull + csize + bsum;
				x->frames[fi].n_blocks++;
			}
			if (end >= 0) {
				if ((!at_eof || budget_stop) && rs) {
					/* the frame goes on in the next window: its complete blocks are decoded
					 * now, the content hash is carried over */
					x->frames[fi].flags = (x->frames[fi].flags & ~LA_LZ4F_CONTENT_SUM) | LA_LZ4F_OPEN;
					rs->in_frame = 1;
					rs->bmax = bmax;
					rs->flags = (bsum ? 1u : 0u) | (ssum ? 2u : 0u) | (indep ? 0u : 4u);
					rs->blocks_so_far = blocks_before + x->frames[fi].n_blocks;
					x->end_kind = LA_END_NEED_MORE;
					x->consumed = p;
					goto out;
				}
				if (!at_eof) {
					/* incomplete frame in a window: hand the whole frame to the next window */
					x->n_blocks = save_b; x->n_frames = save_f; x->max_out = save_out;
					x->end_kind = LA_END_NEED_MORE;
					goto out;
				}
				/* at EOF: the reference delivers the complete blocks, then fails.
				 * The frame stays indexed without its content checksum. */
				x->frames[fi].flags &= ~LA_LZ4F_CONTENT_SUM;
				x->end_kind = end;
				x->consumed = p;
				goto out;
			}
			pos = p;
			if (x->frames[fi].n_blocks == 0 && blocks_before == 0) {
				x->consumed = pos;
				x->end_kind = LA_END_EMPTY_FRAME;
				goto out;
			}
			continue;
frame_cut:
			/* malformed size word: blocks before it are delivered, then the error;
			 * the content checksum is never reached */
			x->frames[fi].flags &= ~LA_LZ4F_CONTENT_SUM;
			goto out;
		} else if (m == LZ4_LEGACY) {
			/* lz4.c:670-721.  Legacy blocks share nothing (no checksums, no dictionary), so a
			 * legacy frame larger than the window simply continues with the next window. */
			const int cont = resume_legacy;
			const uint32_t blocks_before = cont ? rs->blocks_so_far : 0;
			uint64_t p = cont ? pos : pos + 4;
			if (cont) {
				resume_legacy = 0;
				rs->in_frame = 0;
			}
			uint32_t save_b = x->n_blocks, save_f = x->n_frames;
			uint64_t save_out = x->max_out;
			la_lz4_frame *f = push_frame(x);
			if (!f) return -1;
			uint32_t fi = x->n_frames - 1;
			f->first_block = x->n_blocks;
			int end = -1, budget_stop = 0;
			for (;;) {
				if (len - p < 4) {
					if (x->frames[fi].n_blocks + blocks_before == 0 || !at_eof)
						end = LA_END_TRUNCATED;	/* lz4.c:685-692 (first block) */
					break;
				}
				uint32_t csize = le32(img + p);
				if (csize > LEGACY_BOUND)
					break;	/* not a block: re-read as a magic number (lz4.c:698-701) */
				if (len - p < 4ull + csize) { end = LA_END_TRUNCATED; break; }
				if (out_budget && rs && x->frames[fi].n_blocks > 0 && x->max_out + LEGACY_BLOCK > out_budget) {
					end = LA_END_TRUNCATED; budget_stop = 1; break;
				}
				la_lz4_block *b = push_block(x);
				if (!b) return -1;
				b->src_off = p + 4;
				b->src_len = csize;
				b->dst_cap = LEGACY_BLOCK;
				b->flags = 0;
				b->block_sum = 0;
				x->max_out += LEGACY_BLOCK;
				p += 4ull + csize;
				x->frames[fi].n_blocks++;
			}
			if (end >= 0) {
				if ((!at_eof || budget_stop) && rs) {
					/* the complete blocks are decoded now, the frame goes on in the next window */
					rs->in_frame = 2;
					rs->blocks_so_far = blocks_before + x->frames[fi].n_blocks;
					x->end_kind = LA_END_NEED_MORE;
					x->consumed = p;
					goto out;
				}
				if (!at_eof) {
					x->n_blocks = save_b; x->n_frames = save_f; x->max_out = save_out;
					x->end_kind = LA_END_NEED_MORE;
					goto out;
				}
				x->end_kind = end;
				x->consumed = p;
				goto out;
			}
			pos = p;
			if (x->frames[fi].n_blocks + blocks_before == 0) {
				/* legacy magic without a block: the read() that selected it returns 0 */
				x->consumed = pos;
				x->end_kind = LA_END_EOF;
				goto out;
			}
			continue;
		} else if ((m & ~0xFu) == LZ4_SKIP) {
			if (len - pos < 8)
				SHORT(LA_END_MALFORMED_SKIP);
			uint64_t skip = 8ull + le32(img + pos + 4);
			if (skip > len - pos) {
				if (!at_eof) { x->end_kind = LA_END_NEED_MORE; goto out; }
				pos = len;	/* unchecked consume past the end, then ahead() sees EOF */
			} else
				pos += skip;
			continue;
		} else {
			/* lz4.c:358-363: unrecognised data ends the stream silently */
			x->end_kind = LA_END_EOF;
			goto out;
		}
	}
out:
	return 0;
}

const char *la_status_message(uint32_t st)
{
	switch (st) {
	case LA_ST_OK: return "";
	case LA_ST_LZ4_BAD_BLOCK_SUM:
	case LA_ST_LZ4_BAD_HEADER_SUM: return "malformed lz4 data";
	case LA_ST_LZ4_DECODE: return "lz4 decompression failed";
	case LA_ST_LZ4_BAD_CONTENT_SUM: return "lz4 stream checksum error";
	case LA_ST_GZ_DATA: return "gzip decompression failed";
	case LA_ST_GZ_TRUNCATED: return "truncated gzip input";
	case LA_ST_GZ_BAD_CRC: return "gzip member CRC32 mismatch";
	case LA_ST_GZ_BAD_ISIZE: return "gzip member ISIZE mismatch";
	/* zstd.c:226-231 "Zstd decompression failed: %s" with libzstd's ZSTD_getErrorName() text */
	case LA_ST_ZSTD_CORRUPT: return "Zstd decompression failed: Corrupted block detected";
	case LA_ST_ZSTD_TRUNCATED: return "Truncated zstd input";
	case LA_ST_ZSTD_BAD_CHECKSUM: return "Zstd decompression failed: Restored data doesn't match checksum";
	case LA_ST_ZSTD_OUT_FULL: return "Zstd decompression failed: Corrupted block detected";
	case LA_ST_ZSTD_UNSUPPORTED: return "Zstd decompression failed: Unsupported frame parameter";
	case LA_ST_ZSTD_WINDOW: return "Zstd decompression failed: Frame requires too much memory for decoding";
	case LA_ST_ZSTD_DICTIONARY: return "Zstd decompression failed: Dictionary mismatch";
	default: return "unknown device status";
	}
}

const char *la_end_message(int end_kind, int is_gzip)
{
	switch (end_kind) {
	case LA_END_TRUNCATED: return is_gzip ? "truncated gzip input" : "truncated lz4 input";
	case LA_END_MALFORMED: return "malformed lz4 data";
	case LA_END_MALFORMED_SKIP: return "Malformed lz4 data";
	case LA_END_GZ_NO_TRAILER: return "";
	case LA_END_GZ_TOO_LARGE: return "gzip member too large for the GPU data plane (4 GiB limit)";
	default: return "";
	}
}
