/*
 * la_filter_zstd.c -- the zstd read filter on the MI355X data plane.
 *
 * Mirrors libarchive/archive_read_support_filter_zstd.c: the same bidder (zstd.c:107-131: 32 bits on the frame
 * magic or on a skippable-frame magic), filter code and name (ARCHIVE_FILTER_ZSTD, "zstd"), vtable shape
 * (read / close) and error strings ("Truncated zstd input", zstd.c:213-217; "Zstd decompression failed: %s" with
 * libzstd's error names, zstd.c:226-231).  Where the reference feeds ZSTD_decompressStream whatever
 * __archive_read_filter_ahead returns and hands out one ZSTD_DStreamOutSize() buffer per read, this filter gathers
 * a window of WHOLE frames (la_zstd_index_build: frame and block headers only), decodes them in one
 * la_gpu_zstd_decode call -- one frame per lane -- and hands out one frame per read.  Bytes in front of an error are
 * delivered first, as the reference's loop does.  There is no CPU fallback.
 */
#include "la_read_private.h"
#include "la_host.h"
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct zstd_private {
	la_gpu_ctx *gpu;
	size_t batch_bytes;		/* compressed bytes gathered per window */
	size_t target_bytes;		/* the window ramps up to this size (16, 32, 64 MiB ...: la_filter_lz4.c, init) */
	uint64_t out_budget;		/* decoded bytes asked for per window */
	size_t max_batch_bytes;		/* how far the stage may grow for ONE frame larger than a window */
	/* stage: compressed bytes not decoded yet */
	uint8_t *stage; size_t stage_len, stage_cap;
	int upstream_eof;
	uint64_t skip_left;		/* bytes of a skippable frame still to pass over before anything is staged (zstd.c skips
					 * such frames in constant memory: so does this filter) */
	/* Two windows in flight (round 3, as the lz4 filter does): while the caller consumes the slab of window n, window
	 * n + 1 is already gathered, indexed and queued on the device (H2D, decode, D2H into the OTHER slot's pinned slab);
	 * only that slab copy is waited for (la_gpu_mark / la_gpu_wait_mark).  The device buffers are shared: the stream
	 * orders window n's slab copy in front of window n + 1's decode.  What a window found (frames, results, what
	 * follows its last frame, a refusal) stays with its slot and is reported when the slot's turn comes: bytes in
	 * front of an error are delivered first, as the reference's loop does. */
	struct zstd_slot {
		la_zstd_frame *frames; uint32_t frames_cap;
		la_zstd_result *results;
		uint8_t *out; size_t out_cap;		/* pinned */
		uint32_t n, next;			/* frames of the window / next to hand out */
		int end_kind;				/* what follows the window's last frame */
		int launched;				/* the window has been prepared (and queued when n > 0) */
		int rc;					/* ARCHIVE_OK, or what preparing it ended with (reported at its turn) */
		char err[200];
	} slot[2];
	int cur;				/* the slot being handed out */
	void *d_src, *d_dst, *d_frames, *d_results;
	size_t d_src_cap, d_dst_cap, d_tab_cap;
	int finished;				/* error or end already reported */
	int64_t total_out;
};

static int zstd_reader_bid(struct archive_read_filter_bidder *, struct archive_read_filter *);
static int zstd_reader_init(struct archive_read_filter *);
static ssize_t zstd_filter_read(struct archive_read_filter *, const void **);
static int zstd_filter_close(struct archive_read_filter *);

static const struct archive_read_filter_bidder_vtable zstd_bidder_vtable = {
	.bid = zstd_reader_bid,
	.init = zstd_reader_init,
};
static const struct archive_read_filter_vtable zstd_reader_vtable = {
	.read = zstd_filter_read,
	.close = zstd_filter_close,
};

int archive_read_support_filter_zstd(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	if (__archive_read_register_bidder(a, NULL, "zstd", &zstd_bidder_vtable) != ARCHIVE_OK)
		return ARCHIVE_FATAL;
	return ARCHIVE_OK;
}

static int zstd_reader_bid(struct archive_read_filter_bidder *self, struct archive_read_filter *filter)
{
	ssize_t avail;
	(void)self;
	const unsigned char *p = __archive_read_filter_ahead(filter, 4, &avail);
	if (p == NULL)
		return 0;
	const int bits = la_zstd_bid_bytes(p, 4);
	/* Default policy (la_bid_policy.c): a frame is one serial chain (one wave); a stream whose first frame does not
	 * end inside the look-ahead is left to libzstd on a host core. */
	if (bits > 0 && !la_bid_take_all()) {
		const size_t la = la_bid_lookahead(1024);
		size_t got = 0;
		const unsigned char *w = la_bid_peek(filter, la, &got);
		if (w != NULL && !la_bid_zstd_parallel(w, got, la))
			return 0;
	}
	return bits;
}

static int zstd_reader_init(struct archive_read_filter *self)
{
	self->code = ARCHIVE_FILTER_ZSTD;
	self->name = "zstd";
	struct zstd_private *st = calloc(1, sizeof(*st));
	if (st == NULL) {
		archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for zstd decompression");
		return ARCHIVE_FATAL;
	}
	const char *dev = getenv("LA_GPU_DEVICE");
	const char *bm = getenv("LA_GPU_BATCH_MIB");
	st->target_bytes = (size_t)(bm && atoi(bm) > 0 ? atoi(bm) : 64) << 20;
	st->batch_bytes = st->target_bytes < ((size_t)16 << 20) ? st->target_bytes : (size_t)16 << 20;
	const char *ob = getenv("LA_GPU_OUT_BUDGET_MIB");
	st->out_budget = (uint64_t)(ob && atoi(ob) > 0 ? atoi(ob) : 4096) << 20;
	const char *bmx = getenv("LA_GPU_MAX_BATCH_MIB");
	st->max_batch_bytes = (size_t)(bmx && atoi(bmx) > 0 ? atoi(bmx) : 2048) << 20;
	int rc = la_gpu_open(dev ? atoi(dev) : 0, &st->gpu);
	if (rc != LA_OK) {
		archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC,
		    "Can't initialize zstd GPU data plane (la_gpu_open: %d); no CPU fallback is built", rc);
		free(st);
		return ARCHIVE_FATAL;
	}
	self->data = st;
	self->vtable = &zstd_reader_vtable;
	return ARCHIVE_OK;
}

static int gpu_fail(struct archive_read_filter *self, struct zstd_private *st, const char *what)
{
	archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC,
	    "zstd GPU data plane: %s failed: %s", what, la_gpu_last_error(st->gpu));
	st->finished = 1;
	return ARCHIVE_FATAL;
}

static int grow_dev(struct zstd_private *st, void **p, size_t *cap, size_t need)
{
	if (need <= *cap)
		return LA_OK;
	if (*p)
		la_gpu_free(st->gpu, *p);
	*p = NULL;
	*cap = 0;
	need = (need + (need >> 2) + 0xFFFFFu) & ~(size_t)0xFFFFFu;
	int rc = la_gpu_malloc(st->gpu, p, need);
	if (rc == LA_OK)
		*cap = need;
	return rc;
}

/* Pull upstream into the stage until it holds a window's worth (or upstream ends). */
static int zstd_fill(struct archive_read_filter *self, struct zstd_private *st, size_t want)
{
	while (!st->upstream_eof && st->stage_len < want) {
		ssize_t avail;
		const void *up = __archive_read_filter_ahead(self->upstream, 1, &avail);
		if (up == NULL) {
			if (avail < 0)
				return (int)avail;
			st->upstream_eof = 1;
			break;
		}
		if (st->skip_left) {
			/* the rest of a skippable frame: consumed, never copied */
			const uint64_t k = st->skip_left < (uint64_t)avail ? st->skip_left : (uint64_t)avail;
			__archive_read_filter_consume(self->upstream, (int64_t)k);
			st->skip_left -= k;
			continue;
		}
		if (st->stage_len >= 8 && st->stage_len < 8 + (uint64_t)0xFFFFFFFFu) {
			/* a stage that BEGINS with a skippable frame (0x184D2A5x + 32-bit size, zstd.c:107-131 / RFC 8878 3.1.2):
			 * drop what is there of it and remember how much is still to come */
			const uint32_t mg = (uint32_t)st->stage[0] | (uint32_t)st->stage[1] << 8 | (uint32_t)st->stage[2] << 16 | (uint32_t)st->stage[3] << 24;
			if ((mg & 0xFFFFFFF0u) == 0x184D2A50u) {
				const uint64_t total = 8ull + ((uint64_t)st->stage[4] | (uint64_t)st->stage[5] << 8 | (uint64_t)st->stage[6] << 16 | (uint64_t)st->stage[7] << 24);
				if (total <= st->stage_len) {
					memmove(st->stage, st->stage + total, st->stage_len - (size_t)total);
					st->stage_len -= (size_t)total;
				} else {
					st->skip_left = total - st->stage_len;
					st->stage_len = 0;
				}
				continue;
			}
		}
		size_t n = (size_t)avail;
		if (n > want - st->stage_len)
			n = want - st->stage_len;
		if (st->stage_len + n > st->stage_cap) {
			size_t nc = st->stage_cap ? st->stage_cap : (1u << 20);
			while (nc < st->stage_len + n)
				nc *= 2;
			uint8_t *np = realloc(st->stage, nc);
			if (np == NULL) {
				archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for zstd decompression");
				return ARCHIVE_FATAL;
			}
			st->stage = np;
			st->stage_cap = nc;
		}
		memcpy(st->stage + st->stage_len, up, n);
		st->stage_len += n;
		__archive_read_filter_consume(self->upstream, (int64_t)n);
	}
	return ARCHIVE_OK;
}

/* deferred failure of a window: kept with the slot, reported when the slot's turn comes */
static int slot_fail(struct zstd_slot *sl, int rc, const char *fmt, unsigned long long v)
{
	sl->rc = rc;
	snprintf(sl->err, sizeof(sl->err), fmt, v);
	sl->n = 0;
	sl->end_kind = LA_END_EOF;
	return ARCHIVE_OK;
}

/* Gather, index and QUEUE one window into slot sl (H2D, decode, D2H, marker): nothing is waited for.  Returns
 * ARCHIVE_OK with sl->launched set (sl->n frames, possibly 0; a refusal is kept in sl->rc) or an upstream error. */
static int zstd_launch(struct archive_read_filter *self, struct zstd_private *st, struct zstd_slot *sl)
{
	la_zstd_index_result ir;
	size_t want = st->batch_bytes;
	sl->launched = 1;
	sl->rc = ARCHIVE_OK;
	sl->err[0] = 0;
	sl->n = sl->next = 0;
	sl->end_kind = LA_END_EOF;
	for (;;) {
		int r = zstd_fill(self, st, want);
		if (r != ARCHIVE_OK)
			return r;
		if (sl->frames_cap == 0) {
			sl->frames_cap = 1u << 16;
			sl->frames = malloc(sizeof(la_zstd_frame) * sl->frames_cap);
			sl->results = malloc(sizeof(la_zstd_result) * sl->frames_cap);
			if (!sl->frames || !sl->results) {
				archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for zstd decompression");
				return ARCHIVE_FATAL;
			}
		}
		la_zstd_index_build(st->stage, st->stage_len, st->upstream_eof, st->out_budget, sl->frames, sl->frames_cap, &ir);
		if (ir.n_frames == 0 && ir.end_kind == LA_END_NEED_MORE && !ir.window_full && !st->upstream_eof && ir.consumed > 0) {
			/* nothing but skippable frames in front of an incomplete frame: they are done with -- drop them and gather
			 * on in the same window (the reference skips such frames in constant memory, zstd.c:196-260; growing the
			 * window for them ended in "frame too large" once they passed LA_GPU_MAX_BATCH_MIB) */
			memmove(st->stage, st->stage + ir.consumed, st->stage_len - (size_t)ir.consumed);
			st->stage_len -= (size_t)ir.consumed;
			continue;
		}
		if (ir.n_frames == 0 && ir.end_kind == LA_END_NEED_MORE && !ir.window_full && !st->upstream_eof) {
			if (st->stage_len >= st->max_batch_bytes)
				/* ONE frame whose compressed bytes alone pass LA_GPU_MAX_BATCH_MIB: the whole frame would have to sit in
				 * host memory and HBM; refused by name (the reference streams it) */
				return slot_fail(sl, ARCHIVE_FATAL,
				    "zstd frame too large for the GPU data plane (more than %llu compressed bytes; LA_GPU_MAX_BATCH_MIB)",
				    (unsigned long long)st->max_batch_bytes);
			want = st->stage_len * 2 > want ? st->stage_len * 2 : want * 2;	/* one frame larger than the window: gather on */
			if (want > st->max_batch_bytes)
				want = st->max_batch_bytes;
			continue;
		}
		break;
	}
	sl->n = ir.n_frames;
	sl->end_kind = ir.end_kind;
	if (st->batch_bytes < st->target_bytes)
		st->batch_bytes = st->batch_bytes * 2 < st->target_bytes ? st->batch_bytes * 2 : st->target_bytes;
	if (st->out_budget && ir.dst_bytes > st->out_budget && ir.dst_bytes > ((uint64_t)4 << 30))
		/* ONE frame whose blocks may decode to more than the window's budget (the walker stops adding frames at the
		 * budget, so this is a single frame: e.g. terabytes of one byte as RLE blocks).  The reference streams such a
		 * frame 128 KiB at a time; this data plane decodes whole frames into HBM and refuses it by name. */
		return slot_fail(sl, ARCHIVE_FATAL,
		    "zstd frame too large for the GPU data plane (its blocks may decode to %llu bytes; LA_GPU_OUT_BUDGET_MIB)",
		    (unsigned long long)ir.dst_bytes);
	if (sl->n) {
		const size_t tab = sizeof(la_zstd_frame) * sl->n, rtab = sizeof(la_zstd_result) * sl->n;
		if (grow_dev(st, &st->d_src, &st->d_src_cap, (size_t)ir.consumed + 64) != LA_OK) return gpu_fail(self, st, "la_gpu_malloc");
		if (grow_dev(st, &st->d_dst, &st->d_dst_cap, (size_t)ir.dst_bytes + 64) != LA_OK) return gpu_fail(self, st, "la_gpu_malloc");
		if (tab + rtab > st->d_tab_cap) {
			if (st->d_frames) la_gpu_free(st->gpu, st->d_frames);
			st->d_frames = NULL;
			st->d_tab_cap = 0;
			if (la_gpu_malloc(st->gpu, &st->d_frames, 2 * (tab + rtab)) != LA_OK) return gpu_fail(self, st, "la_gpu_malloc");
			st->d_tab_cap = 2 * (tab + rtab);
		}
		st->d_results = (uint8_t *)st->d_frames + ((tab + 15u) & ~(size_t)15u);
		if ((size_t)ir.dst_bytes > sl->out_cap) {
			if (sl->out) la_gpu_free_host(st->gpu, sl->out);
			sl->out = NULL;
			sl->out_cap = 0;
			void *hp = NULL;
			const size_t nc = ((size_t)ir.dst_bytes + ((size_t)ir.dst_bytes >> 2) + 0xFFFFFu) & ~(size_t)0xFFFFFu;
			if (la_gpu_malloc_host(st->gpu, &hp, nc) != LA_OK) return gpu_fail(self, st, "la_gpu_malloc_host");
			sl->out = hp;
			sl->out_cap = nc;
		}
		if (la_gpu_memcpy_h2d(st->gpu, st->d_src, st->stage, ir.consumed) != LA_OK) return gpu_fail(self, st, "la_gpu_memcpy_h2d");
		if (la_gpu_memcpy_h2d(st->gpu, st->d_frames, sl->frames, tab) != LA_OK) return gpu_fail(self, st, "la_gpu_memcpy_h2d");
		la_zstd_batch bt;
		memset(&bt, 0, sizeof(bt));
		bt.d_src = st->d_src; bt.src_bytes = ir.consumed;
		bt.d_frames = st->d_frames; bt.n_frames = sl->n;
		bt.d_dst = st->d_dst; bt.dst_cap = ir.dst_bytes;
		bt.d_results = st->d_results;
		{ const char *lk = getenv("LA_ZSTD_LANE_KERNEL"); bt.options = (lk && atoi(lk) > 0) ? LA_ZSTD_OPT_LANE_KERNEL : 0u; }
		if (la_gpu_zstd_decode(st->gpu, &bt) != LA_OK) return gpu_fail(self, st, "la_gpu_zstd_decode");
		if (la_gpu_memcpy_d2h(st->gpu, sl->results, st->d_results, rtab) != LA_OK) return gpu_fail(self, st, "la_gpu_memcpy_d2h");
		if (ir.dst_bytes && la_gpu_memcpy_d2h(st->gpu, sl->out, st->d_dst, ir.dst_bytes) != LA_OK) return gpu_fail(self, st, "la_gpu_memcpy_d2h");
		/* the stage (pageable memory) and the host frame table have been read by the copies above when they return:
		 * the marker covers the rest */
		if (la_gpu_mark(st->gpu) != LA_OK) return gpu_fail(self, st, "la_gpu_mark");
	}
	/* keep what the window did not cover */
	memmove(st->stage, st->stage + ir.consumed, st->stage_len - (size_t)ir.consumed);
	st->stage_len -= (size_t)ir.consumed;
	return ARCHIVE_OK;
}

/* does anything follow the window in slot sl?  (an end kind that ends the stream, or no input left) */
static int slot_is_last(const struct zstd_private *st, const struct zstd_slot *sl)
{
	if (sl->rc != ARCHIVE_OK)
		return 1;
	if (sl->end_kind == LA_END_TRUNCATED || sl->end_kind == LA_END_ZSTD_BAD_MAGIC || sl->end_kind == LA_END_ZSTD_BAD_BLOCK)
		return 1;
	return sl->end_kind == LA_END_EOF && st->upstream_eof && st->stage_len == 0;
}

static ssize_t zstd_filter_read(struct archive_read_filter *self, const void **p)
{
	struct zstd_private *st = (struct zstd_private *)self->data;
	*p = NULL;
	if (st->finished)
		return 0;
	for (;;) {
		struct zstd_slot *sl = &st->slot[st->cur];
		if (!sl->launched) {
			/* the very first window: prepare it, wait for it, and put the second one in flight behind it */
			int r = zstd_launch(self, st, sl);
			if (r != ARCHIVE_OK)
				return r;
			if (sl->n && la_gpu_wait_mark(st->gpu) != LA_OK)
				return gpu_fail(self, st, "la_gpu_wait_mark");
			if (!slot_is_last(st, sl)) {
				r = zstd_launch(self, st, &st->slot[st->cur ^ 1]);
				if (r != ARCHIVE_OK)
					return r;
			}
		}
		while (sl->next < sl->n) {
			const uint32_t i = sl->next++;
			const la_zstd_result *r = &sl->results[i];
			if (r->status != LA_ST_OK) {
				archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC, "%s", la_status_message(r->status));
				st->finished = 1;
				return ARCHIVE_FATAL;
			}
			if (r->out_len == 0)
				continue;	/* (the reference's loop goes on over an empty frame: zstd.c:196-239) */
			/* frames whose output lies back to back in the slab (slots are 16-byte aligned: full slots of a
			 * multiple of 16 bytes) go out in one read */
			uint64_t len = r->out_len;
			while (sl->next < sl->n && sl->results[sl->next].status == LA_ST_OK && sl->results[sl->next].out_len != 0 &&
			    sl->frames[sl->next].dst_off == sl->frames[i].dst_off + len && len < ((uint64_t)1 << 30))
				len += sl->results[sl->next++].out_len;
			*p = sl->out + sl->frames[i].dst_off;
			st->total_out += (int64_t)len;
			return (ssize_t)len;
		}
		/* the window is handed out: what did preparing it end with, and what came behind its last frame? */
		if (sl->rc != ARCHIVE_OK) {
			archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC, "%s", sl->err);
			st->finished = 1;
			return sl->rc;
		}
		{
			const char *m = NULL;
			switch (sl->end_kind) {
			case LA_END_TRUNCATED: m = "Truncated zstd input"; break;	/* zstd.c:213-217 */
			case LA_END_ZSTD_BAD_MAGIC: m = "Zstd decompression failed: Unknown frame descriptor"; break;
			case LA_END_ZSTD_BAD_BLOCK: m = "Zstd decompression failed: Corrupted block detected"; break;	/* (the device reported the frame) */
			default: break;
			}
			if (m) {
				archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC, "%s", m);
				st->finished = 1;
				return ARCHIVE_FATAL;
			}
		}
		struct zstd_slot *nx = &st->slot[st->cur ^ 1];
		if (!nx->launched) {
			/* nothing was put in flight behind this window: it was the last one (zstd.c:208-211: end of input on a
			 * frame boundary), or the walker wants more input than one window gave it */
			if (sl->end_kind == LA_END_EOF && st->upstream_eof && st->stage_len == 0) {
				st->finished = 1;
				return 0;
			}
			int r = zstd_launch(self, st, nx);
			if (r != ARCHIVE_OK)
				return r;
		}
		/* the next window's turn: wait for its slab, then put the one after it in flight into the slot just emptied */
		if (nx->n && la_gpu_wait_mark(st->gpu) != LA_OK)
			return gpu_fail(self, st, "la_gpu_wait_mark");
		st->cur ^= 1;
		sl->launched = 0;
		if (!slot_is_last(st, nx)) {
			int r = zstd_launch(self, st, sl);
			if (r != ARCHIVE_OK)
				return r;
		}
	}
}

static int zstd_filter_close(struct archive_read_filter *self)
{
	struct zstd_private *st = (struct zstd_private *)self->data;
	if (st == NULL)
		return ARCHIVE_OK;
	if (st->gpu) {
		if (st->d_src) la_gpu_free(st->gpu, st->d_src);
		if (st->d_dst) la_gpu_free(st->gpu, st->d_dst);
		if (st->d_frames) la_gpu_free(st->gpu, st->d_frames);
		(void)la_gpu_sync(st->gpu);	/* a window may still be in flight */
		for (int i = 0; i < 2; i++)
			if (st->slot[i].out) la_gpu_free_host(st->gpu, st->slot[i].out);
		la_gpu_close(st->gpu);
	}
	free(st->stage);
	for (int i = 0; i < 2; i++) {
		free(st->slot[i].frames);
		free(st->slot[i].results);
	}
	free(st);
	self->data = NULL;
	return ARCHIVE_OK;
}
