/*
 * la_filter_lz4.c -- the lz4 read filter with an MI355X data plane.
 *
 * Drop-in for libarchive/archive_read_support_filter_lz4.c: same public entry
 * point (archive_read_support_filter_lz4, lz4.c:113-129), same bidder
 * (lz4.c:138-183), same filter name/code, same vtable shape {read, close}
 * (lz4.c:209-213), same return codes and error strings.  What changes is the
 * inside of read(): instead of one LZ4_decompress_safe + XXH32 per call it
 *   1. gathers a large window of the compressed stream from upstream
 *      (__archive_read_filter_ahead / _consume, unchanged contracts),
 *   2. walks the frame / block headers on the host (la_lz4_index_build),
 *   3. ships window + job tables to the GPU (la_gpu_lz4_decode: block
 *      checksums, decode, content checksums, header check bytes),
 *   4. returns the decoded slab (a host buffer that stays valid until the next
 *      read(), as the vtable contract requires -- archive_read_private.h:76-83,
 *      archive_read.c:1413-1415).
 * Errors are reported in STREAM order: blocks before the failing one are
 * delivered by one read(), the following read() returns ARCHIVE_FATAL with the
 * reference's message for that failure.
 *
 * There is no CPU decode path: without a usable GPU, init() fails the open.
 *
 * Environment (read filters take no options, archive_read_set_options.c:110-123):
 *   LA_GPU_DEVICE     device ordinal (default 0)
 *   LA_GPU_BATCH_MIB  compressed bytes gathered per window once the ramp (16, 32, ... MiB) has reached it (default 64)
 *   LA_GPU_MAX_BATCH_MIB  how far a window of few, large blocks may grow (default 2048)
 */
#include "la_read_private.h"
#include "../../include/la_gpu.h"
#include "../../include/la_host.h"
#include <errno.h>
#include <stdio.h>

/* one window in flight: its buffers, its index, what the device said about it */
#define LA_HIST_BYTES 65536u	/* what a dependent block may reach back into (and the headroom in front of every slab) */

struct lz4_slot {
	/* compressed window (pinned host memory) */
	uint8_t *stage;
	size_t stage_cap, stage_len;
	/* device buffers, grown on demand */
	void *d_src, *d_dst, *d_tabs;
	size_t d_src_cap, d_dst_cap, d_tabs_cap;
	/* decoded slab handed to the read core (pinned host memory) */
	uint8_t *slab;
	size_t slab_cap;
	la_batch_summary *h_sum;	/* pinned: the summary lands here asynchronously */
	la_lz4_index idx;
	int have_idx;		/* idx is valid (and owns memory) */
	int launched;		/* device work for idx is queued */
	size_t o_outlen, o_dstoff, o_bst, o_fst;	/* layout of d_tabs */
};

struct lz4_private {
	la_gpu_ctx *gpu;
	/* Two windows: while the caller consumes the slab of one (and while that slab is still
	 * on its way over PCIe), the next window is gathered from upstream, indexed and queued
	 * on the device.  Host copying, H2D, kernels and D2H of neighbouring windows overlap;
	 * the order in which bytes and errors come out of read() does not change. */
	struct lz4_slot slot[2];
	int cur;
	size_t batch_bytes, max_batch_bytes;
	size_t target_bytes;	/* the window ramps up to this size: 16, 32, 64 MiB ... */
	uint64_t out_budget;	/* decoded bytes (sum of block maxima) one window may ask for: bounds d_dst and the pinned slab */
	la_lz4_resume rs;	/* a frame of independent blocks may span windows: where the walker is */
	uint8_t *d_carry;	/* 2 x LA_XXH_CARRY_BYTES on the device: content-hash state from window to window */
	uint8_t *d_hist;	/* 64 KiB on the device: the last block of a window, dictionary of the next one's first
				 * block when a frame of DEPENDENT blocks spans windows (lz4.c:563-577) */
	uint32_t hist_len;
	int carry_flip;		/* which half the next window reads */
	int upstream_eof;
	int upstream_fatal;	/* upstream failed while the NEXT window was gathered: reported after this one */
	/* host copies of per-unit results (only fetched when a batch has an event) */
	uint32_t *h_u32;
	size_t h_u32_cap;
	/* what the NEXT read() must report */
	int pending_fatal;
	char pending_msg[128];
	int eof;
};

static int lz4_reader_bid(struct archive_read_filter_bidder *, struct archive_read_filter *);
static int lz4_reader_init(struct archive_read_filter *);
static ssize_t lz4_filter_read(struct archive_read_filter *, const void **);
static int lz4_filter_close(struct archive_read_filter *);

static const struct archive_read_filter_bidder_vtable lz4_bidder_vtable = {
	.bid = lz4_reader_bid,
	.init = lz4_reader_init,
};

static const struct archive_read_filter_vtable lz4_reader_vtable = {
	.read = lz4_filter_read,
	.close = lz4_filter_close,
};

int archive_read_support_filter_lz4(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	if (__archive_read_register_bidder(a, NULL, "lz4", &lz4_bidder_vtable) != ARCHIVE_OK)
		return ARCHIVE_FATAL;
	return ARCHIVE_OK;
}

/* lz4.c:138-183: needs 11 bytes; 48 bits for a frame header, 32 for legacy */
static int lz4_reader_bid(struct archive_read_filter_bidder *self, struct archive_read_filter *filter)
{
	ssize_t avail;
	(void)self;
	const unsigned char *p = __archive_read_filter_ahead(filter, 11, &avail);
	if (p == NULL)
		return 0;
	const int bits = la_lz4_bid_bytes(p, 11);
	/* Default policy (la_bid_policy.c): a frame WITH a content checksum that does not end inside the look-ahead is
	 * bound by one XXH32 chain (one DPP row, two thirds of a host core's rate): not bid for. */
	if (bits > 0 && !la_bid_take_all()) {
		const size_t la = la_bid_lookahead(1024);
		size_t got = 0;
		const unsigned char *w = la_bid_peek(filter, la, &got);
		if (w != NULL && !la_bid_lz4_parallel(w, got, la))
			return 0;
	}
	return bits;
}

static int gpu_fail(struct archive_read_filter *self, struct lz4_private *st, const char *what)
{
	archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC,
	    "lz4 GPU data plane: %s failed: %s", what, st->gpu ? la_gpu_last_error(st->gpu) : "no device");
	return ARCHIVE_FATAL;
}

static int lz4_reader_init(struct archive_read_filter *self)
{
	self->code = ARCHIVE_FILTER_LZ4;
	self->name = "lz4";

	struct lz4_private *st = calloc(1, sizeof(*st));
	if (st == NULL) {
		archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for lz4 decompression");
		return ARCHIVE_FATAL;
	}
	const char *dev = getenv("LA_GPU_DEVICE");
	const char *bm = getenv("LA_GPU_BATCH_MIB");
	/* The window RAMPS: the first one holds 16 MiB of the stream, the next 32, up to the target.  What a window
	 * costs before its first byte comes back -- pinned staging and slab of its size (about half a millisecond
	 * per MiB), the gather, the upload -- is paid before anything overlaps, so a stream of 1 GiB took 1.0 s
	 * with fixed 256 MiB windows and 0.48 s with 16 MiB ones, and 16 GiB 1.70 s against 1.41 s at 64 MiB
	 * (profiles/r03_alevel.txt): small streams want small windows, long ones 64-128 MiB. */
	st->target_bytes = (size_t)(bm && atoi(bm) > 0 ? atoi(bm) : 64) << 20;
	st->batch_bytes = st->target_bytes < ((size_t)16 << 20) ? st->target_bytes : (size_t)16 << 20;
	const char *bmx = getenv("LA_GPU_MAX_BATCH_MIB");
	st->max_batch_bytes = (size_t)(bmx && atoi(bmx) > 0 ? atoi(bmx) : 2048) << 20;
	const char *ob = getenv("LA_GPU_OUT_BUDGET_MIB");
	st->out_budget = (uint64_t)(ob && atoi(ob) > 0 ? atoi(ob) : 4096) << 20;
	int rc = la_gpu_open(dev ? atoi(dev) : 0, &st->gpu);
	if (rc != LA_OK) {
		archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC,
		    "Can't initialize lz4 GPU data plane (la_gpu_open: %d); no CPU fallback is built", rc);
		free(st);
		return ARCHIVE_FATAL;
	}
	void *cp = NULL;
	if (la_gpu_malloc(st->gpu, &cp, 2 * LA_XXH_CARRY_BYTES + LA_HIST_BYTES) != LA_OK) {
		archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC, "Can't allocate lz4 GPU state");
		la_gpu_close(st->gpu);
		free(st);
		return ARCHIVE_FATAL;
	}
	st->d_carry = cp;
	st->d_hist = (uint8_t *)cp + 2 * LA_XXH_CARRY_BYTES;
	self->data = st;
	self->vtable = &lz4_reader_vtable;
	return ARCHIVE_OK;
}

static int grow_pinned(struct lz4_private *st, uint8_t **p, size_t *cap, size_t need, size_t keep)
{
	if (*cap >= need)
		return 0;
	size_t nc = *cap ? *cap : (1u << 20);
	while (nc < need)
		nc *= 2;
	void *np = NULL;
	if (la_gpu_malloc_host(st->gpu, &np, nc) != LA_OK)
		return -1;
	if (keep)
		memcpy(np, *p, keep);
	if (*p)
		la_gpu_free_host(st->gpu, *p);
	*p = np;
	*cap = nc;
	return 0;
}

static int grow_dev(struct lz4_private *st, void **p, size_t *cap, size_t need)
{
	if (*cap >= need)
		return 0;
	size_t nc = *cap ? *cap : (1u << 20);
	while (nc < need)
		nc *= 2;
	if (*p)
		la_gpu_free(st->gpu, *p);
	*p = NULL;
	*cap = 0;
	if (la_gpu_malloc(st->gpu, p, nc) != LA_OK)
		return -1;
	*cap = nc;
	return 0;
}

#define ALIGN256(x) (((x) + 255) & ~(size_t)255)
#define LA_LZ4_MIN_SLOW_BLOCKS 2048u	/* blocks above 64 KiB wanted in one window before it stops growing */

/* what the index alone says about the end of the stream (no device verdict involved) */
static void lz4_apply_end_kind(struct lz4_private *st, int end_kind)
{
	switch (end_kind) {
	case LA_END_TRUNCATED:
	case LA_END_MALFORMED:
	case LA_END_MALFORMED_SKIP:
		st->pending_fatal = 1;
		snprintf(st->pending_msg, sizeof(st->pending_msg), "%s", la_end_message(end_kind, 0));
		break;
	case LA_END_EOF:
	case LA_END_EMPTY_FRAME:
		st->eof = 1;
		break;
	default:	/* LA_END_NEED_MORE: the stream goes on in the next window */
		break;
	}
}

/*
 * Gather one window into sl->stage (which may already hold the carried-over tail of the
 * previous window) and index it.  0 = sl->idx is valid, ARCHIVE_FATAL on error.
 */
static int lz4_gather_and_index(struct archive_read_filter *self, struct lz4_private *st, struct lz4_slot *sl)
{
	for (;;) {
		/* 1. gather compressed bytes until the window is full or upstream ends */
		while (!st->upstream_eof && sl->stage_len < st->batch_bytes) {
			ssize_t avail;
			const void *up = __archive_read_filter_ahead(self->upstream, 1, &avail);
			if (up == NULL) {
				if (avail < 0)
					return ARCHIVE_FATAL;	/* upstream already set the error */
				st->upstream_eof = 1;
				break;
			}
			size_t n = (size_t)avail;
			if (n > st->batch_bytes - sl->stage_len)
				n = st->batch_bytes - sl->stage_len;
			if (grow_pinned(st, &sl->stage, &sl->stage_cap, sl->stage_len + n, sl->stage_len) < 0)
				return gpu_fail(self, st, "pinned staging allocation");
			memcpy(sl->stage + sl->stage_len, up, n);
			sl->stage_len += n;
			__archive_read_filter_consume(self->upstream, (int64_t)n);
		}

		/* 2. frame / block headers (host pointer chase, no payload byte touched) */
		const la_lz4_resume rs_before = st->rs;
		if (la_lz4_index_build3(sl->stage, sl->stage_len, st->upstream_eof, &st->rs, st->out_budget, &sl->idx) != 0) {
			archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for lz4 decompression");
			return ARCHIVE_FATAL;
		}
		if (sl->idx.end_kind == LA_END_NEED_MORE && sl->idx.consumed == 0) {
			/* not one complete item in the window (a block larger than it, or a whole frame of
			 * dependent blocks): widen the window and gather more */
			la_lz4_index_free(&sl->idx);
			st->rs = rs_before;
			st->batch_bytes *= 2;
			continue;
		}
		if (sl->idx.end_kind == LA_END_NEED_MORE && !st->upstream_eof && st->batch_bytes < st->max_batch_bytes &&
		    sl->idx.max_out < st->out_budget / 2) {	/* (a window the decoded-bytes budget cut short does not grow) */
			/* Blocks above 64 KiB are parsed by one lane and expanded by one wave each (about
			 * 0.2 s per 4 MiB block, however many run side by side): their throughput is the
			 * number of blocks in flight, so a window that holds only a few of them grows. */
			uint32_t slow = 0;
			for (uint32_t k = 0; k < sl->idx.n_blocks; k++)
				slow += sl->idx.blocks[k].dst_cap > 65536u;
			if (slow != 0 && slow < LA_LZ4_MIN_SLOW_BLOCKS) {
				la_lz4_index_free(&sl->idx);
				st->rs = rs_before;
				st->batch_bytes *= 2;
				continue;
			}
		}
		sl->have_idx = 1;
		if (st->batch_bytes < st->target_bytes)
			st->batch_bytes = st->batch_bytes * 2 < st->target_bytes ? st->batch_bytes * 2 : st->target_bytes;
		return 0;
	}
}

/* Queue the device work of sl->idx: H2D, decode, summary D2H.  Nothing is waited for. */
static int lz4_launch(struct archive_read_filter *self, struct lz4_private *st, struct lz4_slot *sl)
{
	const la_lz4_index *x = &sl->idx;
	const uint32_t nb = x->n_blocks, nf = x->n_frames;
	const size_t src_len = (size_t)x->consumed;

	sl->launched = 0;
	if (nb == 0 && nf == 0)
		return 0;	/* only skippable frames / trailing bytes in this window: nothing for the device */

	/* layout of the table buffer on the device */
	size_t o = 0;
	const size_t o_blocks = o; o += ALIGN256((size_t)nb * sizeof(la_lz4_block));
	const size_t o_frames = o; o += ALIGN256((size_t)nf * sizeof(la_lz4_frame));
	sl->o_outlen = o; o += ALIGN256((size_t)nb * 4);
	sl->o_dstoff = o; o += ALIGN256(((size_t)nb + 1) * 8);
	sl->o_bst = o; o += ALIGN256((size_t)nb * 4);
	sl->o_fst = o; o += ALIGN256((size_t)nf * 4);
	const size_t o_sum = o; o += 256;
	if (grow_dev(st, &sl->d_src, &sl->d_src_cap, src_len + 64) < 0 ||
	    grow_dev(st, &sl->d_dst, &sl->d_dst_cap, (size_t)x->max_out + 64 + LA_HIST_BYTES) < 0 ||
	    grow_dev(st, &sl->d_tabs, &sl->d_tabs_cap, o) < 0)
		return gpu_fail(self, st, "device allocation");
	if (sl->h_sum == NULL) {
		void *hp = NULL;
		if (la_gpu_malloc_host(st->gpu, &hp, 256) != LA_OK)
			return gpu_fail(self, st, "pinned summary allocation");
		sl->h_sum = hp;
	}
	uint8_t *T = sl->d_tabs;

	if (la_gpu_memcpy_h2d(st->gpu, sl->d_src, sl->stage, src_len) != LA_OK ||
	    la_gpu_memcpy_h2d(st->gpu, T + o_blocks, x->blocks, (size_t)nb * sizeof(la_lz4_block)) != LA_OK ||
	    la_gpu_memcpy_h2d(st->gpu, T + o_frames, x->frames, (size_t)nf * sizeof(la_lz4_frame)) != LA_OK)
		return gpu_fail(self, st, "host to device copy");

	la_lz4_batch bt;
	memset(&bt, 0, sizeof(bt));
	bt.d_src = sl->d_src; bt.src_bytes = src_len;
	bt.d_blocks = (const la_lz4_block *)(T + o_blocks); bt.n_blocks = nb;
	bt.d_frames = nf ? (const la_lz4_frame *)(T + o_frames) : NULL; bt.n_frames = nf;
	bt.d_dst = (uint8_t *)sl->d_dst + LA_HIST_BYTES; bt.dst_cap = x->max_out;	/* (headroom in front: see d_hist) */
	if (nb && (x->blocks[0].flags & LA_LZ4B_HIST)) {
		/* the frame's previous block (at most 64 KiB of it) goes in front of the slab */
		if (la_gpu_memcpy_d2d(st->gpu, bt.d_dst - st->hist_len, st->d_hist, st->hist_len) != LA_OK)
			return gpu_fail(self, st, "history copy");
		bt.hist_len = st->hist_len;
	}
	bt.d_out_len = (uint32_t *)(T + sl->o_outlen);
	bt.d_dst_off = (uint64_t *)(T + sl->o_dstoff);
	bt.d_block_status = (uint32_t *)(T + sl->o_bst);
	bt.d_frame_status = (uint32_t *)(T + sl->o_fst);
	bt.d_summary = (la_batch_summary *)(T + o_sum);
	bt.d_carry_in = st->d_carry + (st->carry_flip ? LA_XXH_CARRY_BYTES : 0);
	bt.d_carry_out = st->d_carry + (st->carry_flip ? 0 : LA_XXH_CARRY_BYTES);
	if (la_gpu_lz4_decode(st->gpu, &bt) != LA_OK)
		return gpu_fail(self, st, "la_gpu_lz4_decode");
	if (nf && (x->frames[nf - 1].flags & LA_LZ4F_OPEN))
		st->carry_flip ^= 1;	/* the next window continues this frame's content hash */
	if (la_gpu_memcpy_d2h(st->gpu, sl->h_sum, T + o_sum, sizeof(*sl->h_sum)) != LA_OK)
		return gpu_fail(self, st, "summary copy");
	sl->launched = 1;
	return 0;
}

/*
 * Wait for the window's verdict, settle what follows its bytes (st->eof / st->pending_fatal)
 * and QUEUE the copy of the bytes to deliver into sl->slab (the caller waits for it with
 * la_gpu_sync).  Returns bytes that will be delivered (>= 0) or ARCHIVE_FATAL.
 */
static ssize_t lz4_resolve(struct archive_read_filter *self, struct lz4_private *st, struct lz4_slot *sl)
{
	const la_lz4_index *x = &sl->idx;
	const uint32_t nb = x->n_blocks, nf = x->n_frames;

	if (!sl->launched) {
		lz4_apply_end_kind(st, x->end_kind);
		return 0;
	}
	if (la_gpu_sync(st->gpu) != LA_OK)
		return gpu_fail(self, st, "summary copy");
	const la_batch_summary sm = *sl->h_sum;
	uint8_t *T = sl->d_tabs;

	uint64_t delivered = sm.total_out;
	uint32_t st_code = LA_ST_OK;
	int silent_end = 0;

	if (sm.n_bad_units || sm.n_bad_frames || sm.first_zero_unit != 0xFFFFFFFFu) {
		/* Something happened: fetch the per-unit words and find the FIRST event in
		 * stream order (per frame: header check -> blocks -> content checksum). */
		size_t words = (size_t)nb * 2 + (size_t)nf + ((size_t)nb + 1) * 2 + 2;
		if (st->h_u32_cap < words) {
			free(st->h_u32);
			st->h_u32 = malloc(words * 4);
			st->h_u32_cap = st->h_u32 ? words : 0;
			if (!st->h_u32) {
				archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for lz4 decompression");
				return ARCHIVE_FATAL;
			}
		}
		uint32_t *h_len = st->h_u32, *h_bst = h_len + nb, *h_fst = h_bst + nb;
		uint64_t *h_off = (uint64_t *)(h_fst + nf + ((nb * 2 + nf) & 1));
		if (la_gpu_memcpy_d2h(st->gpu, h_len, T + sl->o_outlen, (size_t)nb * 4) != LA_OK ||
		    la_gpu_memcpy_d2h(st->gpu, h_bst, T + sl->o_bst, (size_t)nb * 4) != LA_OK ||
		    la_gpu_memcpy_d2h(st->gpu, h_fst, T + sl->o_fst, (size_t)nf * 4) != LA_OK ||
		    la_gpu_memcpy_d2h(st->gpu, h_off, T + sl->o_dstoff, ((size_t)nb + 1) * 8) != LA_OK ||
		    la_gpu_sync(st->gpu) != LA_OK)
			return gpu_fail(self, st, "status copy");
		int found = 0;
		for (uint32_t fi = 0; fi < nf && !found; fi++) {
			const la_lz4_frame *f = &x->frames[fi];
			if (h_fst[fi] == LA_ST_LZ4_BAD_HEADER_SUM) {
				delivered = h_off[f->first_block]; st_code = h_fst[fi]; found = 1;
				break;
			}
			for (uint32_t b = f->first_block; b < f->first_block + f->n_blocks; b++) {
				if (h_bst[b] != LA_ST_OK) {
					delivered = h_off[b]; st_code = h_bst[b]; found = 1;
					break;
				}
				if (h_len[b] == 0) {	/* a block of 0 bytes is EOF to the read core (SURVEY F11 i) */
					delivered = h_off[b]; silent_end = 1; found = 1;
					break;
				}
			}
			if (!found && h_fst[fi] == LA_ST_LZ4_BAD_CONTENT_SUM) {
				delivered = h_off[f->first_block + f->n_blocks]; st_code = h_fst[fi]; found = 1;
			}
		}
	}

	if (st_code != LA_ST_OK) {
		st->pending_fatal = 1;
		snprintf(st->pending_msg, sizeof(st->pending_msg), "%s", la_status_message(st_code));
	} else if (silent_end) {
		st->eof = 1;
	} else {
		lz4_apply_end_kind(st, x->end_kind);
	}

	if (nf && nb && (x->frames[nf - 1].flags & LA_LZ4F_OPEN) && (x->blocks[nb - 1].flags & LA_LZ4B_DEPENDENT) &&
	    x->frames[nf - 1].n_blocks != 0 && !st->pending_fatal && !st->eof) {
		/* the open frame's blocks depend on each other: keep its last block (<= 64 KiB of it)
		 * for the first block of the next window */
		uint32_t last_len = 0;
		if (la_gpu_memcpy_d2h(st->gpu, &last_len, T + sl->o_outlen + (size_t)(nb - 1) * 4, 4) != LA_OK ||
		    la_gpu_sync(st->gpu) != LA_OK)
			return gpu_fail(self, st, "status copy");
		st->hist_len = last_len < LA_HIST_BYTES ? last_len : LA_HIST_BYTES;
		if (la_gpu_memcpy_d2d(st->gpu, st->d_hist, (uint8_t *)sl->d_dst + LA_HIST_BYTES + sm.total_out - st->hist_len,
		    st->hist_len) != LA_OK)
			return gpu_fail(self, st, "history copy");
	}
	if (delivered) {
		if (grow_pinned(st, &sl->slab, &sl->slab_cap, (size_t)delivered, 0) < 0)
			return gpu_fail(self, st, "pinned slab allocation");
		if (la_gpu_memcpy_d2h(st->gpu, sl->slab, (uint8_t *)sl->d_dst + LA_HIST_BYTES, (size_t)delivered) != LA_OK ||
		    la_gpu_mark(st->gpu) != LA_OK)
			return gpu_fail(self, st, "device to host copy");
	}
	return (ssize_t)delivered;
}

/* window `from` has been indexed: its unconsumed tail (an incomplete frame) opens window `to` */
static int lz4_carry_tail(struct archive_read_filter *self, struct lz4_private *st, struct lz4_slot *from,
    struct lz4_slot *to)
{
	const size_t used = (size_t)from->idx.consumed;
	const size_t tail = used < from->stage_len ? from->stage_len - used : 0;
	if (grow_pinned(st, &to->stage, &to->stage_cap, tail ? tail : 1, 0) < 0)
		return gpu_fail(self, st, "pinned staging allocation");
	if (tail)
		memcpy(to->stage, from->stage + used, tail);
	to->stage_len = tail;
	return 0;
}

static void lz4_slot_done(struct lz4_slot *sl)
{
	if (sl->have_idx)
		la_lz4_index_free(&sl->idx);
	sl->have_idx = 0;
	sl->launched = 0;
}

static ssize_t lz4_filter_read(struct archive_read_filter *self, const void **p)
{
	struct lz4_private *st = (struct lz4_private *)self->data;
	*p = NULL;

	for (;;) {
		if (st->pending_fatal) {
			archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC, "%s", st->pending_msg);
			return ARCHIVE_FATAL;
		}
		if (st->eof)
			return 0;
		if (st->upstream_fatal)
			return ARCHIVE_FATAL;	/* upstream set the error while the window was gathered */

		struct lz4_slot *sl = &st->slot[st->cur], *nx = &st->slot[st->cur ^ 1];
		if (!sl->have_idx) {
			/* nothing in flight (first call, or the previous window delivered nothing) */
			int r = lz4_gather_and_index(self, st, sl);
			if (r != 0)
				return r;
			if ((r = lz4_launch(self, st, sl)) != 0)
				return r;
			if ((r = lz4_carry_tail(self, st, sl, nx)) != 0)
				return r;
		}

		/* 3. the window's verdict; its bytes start moving to the host */
		ssize_t n = lz4_resolve(self, st, sl);
		if (n < 0)
			return n;

		/* 4. while they move (and while the caller then consumes them): the next window */
		if (!st->pending_fatal && !st->eof && sl->idx.end_kind == LA_END_NEED_MORE) {
			int r = lz4_gather_and_index(self, st, nx);
			if (r != 0) {
				if (n == 0)
					return r;
				st->upstream_fatal = 1;	/* after this window's bytes */
			} else {
				if ((r = lz4_launch(self, st, nx)) != 0)
					return r;
				if ((r = lz4_carry_tail(self, st, nx, sl)) != 0)	/* (sl's stage is free: its H2D is done) */
					return r;
			}
		}
		/* only the slab copy is waited for: the next window keeps running on the device */
		if (n > 0 && la_gpu_wait_mark(st->gpu) != LA_OK)
			return gpu_fail(self, st, "device to host copy");
		lz4_slot_done(sl);
		st->cur ^= 1;
		if (n > 0) {
			*p = sl->slab;
			return n;
		}
		/* nothing to deliver from this window: report what follows, or go on */
	}
}

static int lz4_filter_close(struct archive_read_filter *self)
{
	struct lz4_private *st = (struct lz4_private *)self->data;
	if (st == NULL)
		return ARCHIVE_OK;
	if (st->gpu) {
		la_gpu_sync(st->gpu);
		for (int i = 0; i < 2; i++) {
			struct lz4_slot *sl = &st->slot[i];
			lz4_slot_done(sl);
			if (sl->stage) la_gpu_free_host(st->gpu, sl->stage);
			if (sl->slab) la_gpu_free_host(st->gpu, sl->slab);
			if (sl->h_sum) la_gpu_free_host(st->gpu, sl->h_sum);
			if (sl->d_src) la_gpu_free(st->gpu, sl->d_src);
			if (sl->d_dst) la_gpu_free(st->gpu, sl->d_dst);
			if (sl->d_tabs) la_gpu_free(st->gpu, sl->d_tabs);
		}
		if (st->d_carry) la_gpu_free(st->gpu, st->d_carry);
		la_gpu_close(st->gpu);
	}
	free(st->h_u32);
	free(st);
	self->data = NULL;
	return ARCHIVE_OK;
}
