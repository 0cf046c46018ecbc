/*
 * la_filter_gzip.c -- the gzip read filter with an MI355X data plane.
 *
 * Drop-in for libarchive/archive_read_support_filter_gzip.c: same public entry
 * points (archive_read_support_filter_gzip and the deprecated
 * archive_read_support_compression_gzip, gzip.c:85-117), same bidder
 * (gzip.c:244-255), same filter name/code, same vtable shape
 * {read, close, read_header} (gzip.c:298-305), same return codes and error
 * strings.  read() gathers a window of the compressed stream, finds the
 * member boundaries on the host (la_gzip_index.c), inflates all members of
 * the window on the GPU (la_gpu_gzip_decode) and returns the decoded slab.
 *
 * Parity details kept on purpose:
 *   - the trailer CRC32/ISIZE is computed on the device but, like the reference
 *     (gzip.c:423), NOT enforced unless LA_GZIP_STRICT=1;
 *   - on an error the reference has delivered whole 64 KiB output blocks only
 *     (gzip.c:314, :446): this filter never hands out the last partial 64 KiB
 *     of what it has decoded until the bytes after it are known to be fine, so
 *     the bytes delivered before an error are exactly the reference's;
 *   - entry name / mtime come from the last member header parsed while the
 *     first 64 KiB were produced (gzip.c:160-163, :196-201, :280-296).
 * There is no CPU decode path: without a usable GPU, init() fails the open.
 *
 * Environment: LA_GPU_DEVICE, LA_GPU_BATCH_MIB (as for lz4), LA_GZIP_STRICT.
 */
#include "la_read_private.h"
#include "../../include/la_gpu.h"
#include "../../include/la_host.h"
#include <errno.h>
#include <stdio.h>
#include <time.h>

/* LA_GPU_TRACE=1: per-window phase times on stderr (diagnostic) */
static double gz_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define OUT_BLOCK 65536u	/* gzip.c:314 */

struct gzip_private {
	la_gpu_ctx *gpu;
	uint8_t *stage;
	size_t stage_cap, stage_len, batch_bytes;
	size_t target_bytes;	/* the window ramps up to this size (16, 32, 64 MiB ...: la_filter_lz4.c, init) */
	int upstream_eof;
	void *d_src, *d_dst, *d_tabs;
	size_t d_src_cap, d_dst_cap, d_tabs_cap;
	uint8_t *slab;		/* [carry | this batch's bytes] */
	size_t slab_cap;
	/* The second slab: while the caller holds `slab`, the NEXT window's decoded bytes are already on their way into this
	 * one (queued behind the decode by gz_prepare, at the offset the carry will take).  When the window turns out as
	 * its index promised -- every member exactly as long as its ISIZE said, nothing refused -- the bytes are in place
	 * when the next read() looks, the carry is put in front and the two slabs change roles; otherwise the copy is done
	 * again the ordinary way.  (The lz4 and zstd filters have two whole slots; here one slab more is what was missing.) */
	uint8_t *slab2;
	size_t slab2_cap;
	int no_ahead;		/* LA_GZ_NO_COPY_AHEAD=1 (measurements: the single-slab behaviour of round 2) */
	int ahead_ok;		/* a copy into slab2 is queued ... */
	size_t ahead_rem;	/* ... behind this many carry bytes ... */
	size_t ahead_len;	/* ... this long */
	size_t carry_len;	/* decoded but not yet delivered (< 64 KiB) */
	size_t last_ret;	/* bytes handed out by the previous read() */
	la_gz_result *h_res;
	size_t h_res_cap;
	uint64_t total_out;	/* bytes decoded so far (delivered + carry) */
	uint32_t hint_skip, hint_cap;
	uint64_t out_budget;	/* decoded bytes (output slots) one window may ask for: LA_GPU_OUT_BUDGET_MIB, default 4096 */
	int strict;
	uint32_t slot_limit;	/* an output slot cannot pass 2 GiB (32-bit positions on the device); LA_GZ_TEST_SLOT_LIMIT lowers it for tests */
	int trace;
	int loose;		/* the stream's headers carry unusual XFL / OS bytes: index without LA_GZ_INDEX_STRICT */
	/* header metadata (gzip.c:280-296) */
	uint32_t mtime;
	char *name;
	/* the window whose device work is queued but not yet looked at (decode-ahead: it was gathered,
	 * uploaded, indexed and launched BEFORE the previous read() returned, so the device works on it
	 * while the caller consumes the slab it was given) */
	la_gz_index idx;
	int inflight;
	size_t o_res;
	int upstream_failed;	/* upstream reported an error while the next window was gathered ahead */
	int pending_fatal;
	int pending_has_msg;
	char pending_msg[128];
	int eof;
};

static int gzip_bidder_bid(struct archive_read_filter_bidder *, struct archive_read_filter *);
static int gzip_bidder_init(struct archive_read_filter *);
static ssize_t gzip_filter_read(struct archive_read_filter *, const void **);
static int gzip_filter_close(struct archive_read_filter *);
static int gzip_read_header(struct archive_read_filter *, struct archive_entry *);

static const struct archive_read_filter_bidder_vtable gzip_bidder_vtable = {
	.bid = gzip_bidder_bid,
	.init = gzip_bidder_init,
};

static const struct archive_read_filter_vtable gzip_reader_vtable = {
	.read = gzip_filter_read,
	.close = gzip_filter_close,
	.read_header = gzip_read_header,
};

int archive_read_support_filter_gzip(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	if (__archive_read_register_bidder(a, NULL, "gzip", &gzip_bidder_vtable) != ARCHIVE_OK)
		return ARCHIVE_FATAL;
	return ARCHIVE_OK;
}

int archive_read_support_compression_gzip(struct archive *a)
{
	return archive_read_support_filter_gzip(a);
}

/*
 * gzip.c:128-239 through the peek interface: the fixed 10 bytes first, then as
 * much as the optional fields need (file names are limited to what upstream
 * can expose in one peek, as in the reference).
 */
static int gzip_bidder_bid(struct archive_read_filter_bidder *self, struct archive_read_filter *filter)
{
	ssize_t avail;
	(void)self;
	const unsigned char *p = __archive_read_filter_ahead(filter, 10, &avail);
	if (p == NULL || avail == 0)
		return 0;
	if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 0x08 || (p[3] & 0xE0))
		return 0;
	/* optional fields: like the reference's parser (gzip.c:183-194), extend the peek one
	 * byte at a time until the header parses or upstream cannot supply another byte */
	for (;;) {
		la_gz_header h;
		if (la_gz_header_parse(p, (size_t)avail, &h)) {
			/* Deployment switch: with LA_GZIP_BID_ONLY_INDEXED=1 this bidder takes only
			 * streams whose first member carries the BGZF "BC" size subfield -- the many-member shape the
			 * device path is built for -- and bids 0 on anything else, so that the reference's own gzip
			 * bidder, registered beside it, wins ordinary single-member .gz files (one serial deflate
			 * chain decodes faster on a host core than on one GPU wave). */
			const char *only = getenv("LA_GZIP_BID_ONLY_INDEXED");
			if (only && only[0] == '1' && h.bgzf_size == 0)
				return 0;
			/* Default policy (la_bid_policy.c): ONE large member is a single serial deflate chain -- one wave on the
			 * device, far slower than zlib on a host core -- so a stream whose first member carries no size subfield and
			 * shows no second member header inside the look-ahead is not bid for (LA_GPU_BID=all takes everything). */
			if (h.bgzf_size == 0 && !la_bid_take_all()) {
				const size_t la = la_bid_lookahead(256);
				size_t got = 0;
				const unsigned char *w = la_bid_peek(filter, la, &got);
				if (w != NULL && !la_bid_gzip_parallel(w, got, h.len, la))
					return 0;
			}
			return 27;
		}
		p = __archive_read_filter_ahead(filter, (size_t)avail + 1, &avail);
		if (p == NULL)
			return 0;
	}
}

static int gz_gpu_fail(struct archive_read_filter *self, struct gzip_private *st, const char *what)
{
	archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC,
	    "gzip GPU data plane: %s failed: %s", what, st->gpu ? la_gpu_last_error(st->gpu) : "no device");
	return ARCHIVE_FATAL;
}

static int gzip_bidder_init(struct archive_read_filter *self)
{
	self->code = ARCHIVE_FILTER_GZIP;
	self->name = "gzip";
	struct gzip_private *st = calloc(1, sizeof(*st));
	if (st == NULL) {
		archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for gzip decompression");
		return ARCHIVE_FATAL;
	}
	const char *dev = getenv("LA_GPU_DEVICE"), *bm = getenv("LA_GPU_BATCH_MIB"), *sv = getenv("LA_GZIP_STRICT");
	st->target_bytes = (size_t)(bm && atoi(bm) > 0 ? atoi(bm) : 64) << 20;
	st->batch_bytes = st->target_bytes < ((size_t)16 << 20) ? st->target_bytes : (size_t)16 << 20;
	st->strict = sv && atoi(sv) != 0;
	st->slot_limit = 0x80000000u;
	{
		const char *sl = getenv("LA_GZ_TEST_SLOT_LIMIT");
		if (sl != NULL && strtoul(sl, NULL, 10) >= 65536 && strtoul(sl, NULL, 10) < st->slot_limit)
			st->slot_limit = (uint32_t)strtoul(sl, NULL, 10);
	}
	const char *ob = getenv("LA_GPU_OUT_BUDGET_MIB");
	st->out_budget = (uint64_t)(ob && atoi(ob) > 0 ? atoi(ob) : 4096) << 20;
	st->trace = getenv("LA_GPU_TRACE") != NULL && atoi(getenv("LA_GPU_TRACE")) != 0;
	st->no_ahead = getenv("LA_GZ_NO_COPY_AHEAD") != NULL && atoi(getenv("LA_GZ_NO_COPY_AHEAD")) != 0;
	int rc = la_gpu_open(dev ? atoi(dev) : 0, &st->gpu);
	if (rc != LA_OK) {
		archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC,
		    "Can't initialize gzip GPU data plane (la_gpu_open: %d); no CPU fallback is built", rc);
		free(st);
		return ARCHIVE_FATAL;
	}
	self->data = st;
	self->vtable = &gzip_reader_vtable;
	return ARCHIVE_OK;
}

static int gzip_read_header(struct archive_read_filter *self, struct archive_entry *entry)
{
	struct gzip_private *st = (struct gzip_private *)self->data;
	if (st->mtime != 0)	/* a mtime of 0 is considered invalid/missing */
		archive_entry_set_mtime(entry, st->mtime, 0);
	if (st->name)
		archive_entry_set_pathname(entry, st->name);
	return ARCHIVE_OK;
}

static int gz_grow_pinned(struct gzip_private *st, uint8_t **p, size_t *cap, size_t need, size_t keep)
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

static int gz_grow_dev(struct gzip_private *st, void **p, size_t *cap, size_t need)
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

static void gz_set_fatal(struct gzip_private *st, const char *msg)
{
	st->pending_fatal = 1;
	st->pending_has_msg = msg != NULL;
	if (msg)
		snprintf(st->pending_msg, sizeof(st->pending_msg), "%s", msg);
}

/*
 * One batch: decode every indexed member, then walk the results in stream
 * order.  On return *used = compressed bytes of the window that are done with,
 * the slab holds carry + newly decoded bytes (st->carry_len updated to the
 * total now waiting), *cutoff = stream offset up to which bytes may be
 * delivered (everything, unless an error follows).
 */
static int gzip_run_batch(struct archive_read_filter *self, struct gzip_private *st,
    const la_gz_index *x, size_t *used, int phase /* 0: queue the device work; 1: results, stream-order walk, slab */)
{
	const uint32_t n = x->n;
	size_t o = 0;
	const size_t o_mem = o; o += ALIGN256((size_t)n * sizeof(la_gz_member));
	const size_t o_res = o; o += ALIGN256((size_t)n * sizeof(la_gz_result));
	const size_t o_sum = o; o += 256;
	size_t src_len = (size_t)x->consumed;
	if (phase == 0 &&
	    (gz_grow_dev(st, &st->d_src, &st->d_src_cap, src_len + 64) < 0 ||
	     gz_grow_dev(st, &st->d_dst, &st->d_dst_cap, (size_t)x->max_out + 64) < 0 ||
	     gz_grow_dev(st, &st->d_tabs, &st->d_tabs_cap, o) < 0))
		return gz_gpu_fail(self, st, "device allocation");
	uint8_t *T = st->d_tabs;
	const double b0 = st->trace ? gz_now() : 0;
	/* (the compressed bytes are already on their way: gz_prepare) */
	if (phase == 0 && la_gpu_memcpy_h2d(st->gpu, T + o_mem, x->members, (size_t)n * sizeof(la_gz_member)) != LA_OK)
		return gz_gpu_fail(self, st, "host to device copy");
	la_gz_batch bt;
	memset(&bt, 0, sizeof(bt));
	bt.d_src = st->d_src; bt.src_bytes = src_len;
	bt.d_members = (const la_gz_member *)(T + o_mem); bt.n_members = n;
	bt.d_dst = st->d_dst; bt.dst_cap = x->max_out;
	bt.d_results = (la_gz_result *)(T + o_res);
	bt.d_summary = (la_batch_summary *)(T + o_sum);
	if (phase == 0) {
		if (la_gpu_gzip_decode(st->gpu, &bt) != LA_OK)
			return gz_gpu_fail(self, st, "la_gpu_gzip_decode");
		return 0;
	}
	if (st->h_res_cap < n) {
		free(st->h_res);
		st->h_res = malloc((size_t)n * sizeof(la_gz_result));
		st->h_res_cap = st->h_res ? n : 0;
		if (!st->h_res) {
			archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for gzip decompression");
			return ARCHIVE_FATAL;
		}
	}
	if (la_gpu_memcpy_d2h(st->gpu, st->h_res, T + o_res, (size_t)n * sizeof(la_gz_result)) != LA_OK ||
	    la_gpu_sync(st->gpu) != LA_OK)
		return gz_gpu_fail(self, st, "result copy");
	const double b1 = st->trace ? gz_now() : 0;

	/* ---- stream-order walk ---- */
	uint64_t total = st->total_out;		/* stream offset of the next decoded byte */
	uint64_t cutoff = UINT64_MAX;		/* deliver only up to here (set when an error follows) */
	uint32_t take = 0;			/* members whose bytes join the slab */
	uint32_t last_out = 0;			/* bytes of a failing member that still count as produced */
	int contiguous = 1;
	int stop = 0;
	*used = src_len;
	const uint32_t prev_skip = st->hint_skip, prev_cap = st->hint_cap;
	st->hint_skip = st->hint_cap = 0;
	for (uint32_t i = 0; i < n && !stop; i++) {
		const la_gz_result *r = &st->h_res[i];
		const la_gz_member *m = &x->members[i];
		const la_gz_header *h = &x->headers[i];
		const uint64_t member_start = h->off;

		/* The deflate stream ran into the end of its span although more input exists:
		 * the boundary (a 1f 8b 08 guess, or a wrong BGZF size) was not the member's
		 * end.  Decode this member again with the span extended past it. */
		if ((r->status == LA_ST_GZ_TRUNCATED &&
		    (m->src_off + m->src_len < st->stage_len || !st->upstream_eof)) ||
		    /* ... or it ended with fewer than 8 bytes left in a span that a BGZF size field or a
		     * boundary guess cut short while the window holds more bytes: same cure */
		    (r->status == LA_ST_GZ_NO_TRAILER && m->src_off + m->src_len < st->stage_len)) {
			*used = (size_t)member_start;
			st->hint_skip = (i == 0 ? prev_skip : 0) + 1;
			st->hint_cap = i == 0 ? prev_cap : 0;
			stop = 1;
			break;
		}
		if (r->status == LA_ST_GZ_OUT_FULL) {
			/* the ISIZE claim was too small for what the member really holds */
			*used = (size_t)member_start;
			uint32_t base = m->dst_cap > prev_cap ? m->dst_cap : prev_cap;
			if (base >= st->slot_limit) {
				/* the slot cannot grow any further (32-bit positions on the device): say so
				 * instead of retrying for ever or delivering a wrapped slot */
				gz_set_fatal(st, la_end_message(LA_END_GZ_TOO_LARGE, 1));
				cutoff = (total / OUT_BLOCK) * OUT_BLOCK;
				stop = 1;
				break;
			}
			st->hint_cap = base < 32768 ? 65536 : (base > st->slot_limit / 2 ? st->slot_limit : base * 2);
			st->hint_skip = i == 0 ? prev_skip : 0;
			stop = 1;
			break;
		}
		if (r->status == LA_ST_GZ_NO_TRAILER && !st->upstream_eof) {
			/* the trailer lies beyond this window */
			*used = (size_t)member_start;
			stop = 1;
			break;
		}
		/* header metadata: parsed by the reference while the first 64 KiB were produced */
		if (total < OUT_BLOCK) {
			st->mtime = h->mtime;
			if (h->name_off) {
				free(st->name);
				st->name = strdup((const char *)st->stage + h->off + h->name_off);
			}
		}
		switch (r->status) {
		case LA_ST_OK:
		case LA_ST_GZ_BAD_CRC:
		case LA_ST_GZ_BAD_ISIZE:
			if (st->strict && r->status != LA_ST_OK) {
				gz_set_fatal(st, la_status_message(r->status));
				cutoff = total;	/* new behaviour: everything before the bad member, then the error */
				stop = 1;
				break;
			}
			if (r->out_len != m->dst_cap)
				contiguous = 0;
			take = i + 1;
			total += r->out_len;
			if (h->bgzf_size && (uint64_t)r->consumed + 8 < m->src_len) {
				/* the member ended before the place its BGZF size field points at: the field is
				 * only a hint (the reference never reads FEXTRA, gzip.c:185-199) and it was wrong.
				 * The stream goes on right behind this member's trailer: index again from there. */
				*used = (size_t)(m->src_off + (uint64_t)r->consumed + 8);
				stop = 1;
				break;
			}
			if (x->speculative && !h->bgzf_size && (uint64_t)r->consumed + 8 < m->src_len) {
				const uint64_t p = m->src_off + (uint64_t)r->consumed + 8;
				const uint8_t *q = st->stage + p;
				const uint64_t rem = st->stage_len - p;
				if (!st->loose && rem >= 4 && q[0] == 0x1f && q[1] == 0x8b && q[2] == 0x08 && (q[3] & 0xE0) == 0) {
					/* a header the strict boundary search passed over (unusual XFL / OS): the
					 * stream goes on here; from now on every 1f 8b 08 is a candidate */
					st->loose = 1;
					*used = (size_t)p;
					stop = 1;
					break;
				}
				/* bytes after the trailer are not a member header: silent end (gzip.c:351-353) */
				st->eof = 1;
				stop = 1;
			}
			break;
		case LA_ST_GZ_DATA:
			gz_set_fatal(st, "gzip decompression failed");
			cutoff = r->out_len == 0 ? (total / OUT_BLOCK) * OUT_BLOCK
			    : ((total + r->out_len - 1) / OUT_BLOCK) * OUT_BLOCK;
			last_out = r->out_len;
			take = i + 1;
			stop = 1;
			break;
		case LA_ST_GZ_TRUNCATED:
			gz_set_fatal(st, "truncated gzip input");
			cutoff = ((total + r->out_len) / OUT_BLOCK) * OUT_BLOCK;
			last_out = r->out_len;
			take = i + 1;
			stop = 1;
			break;
		case LA_ST_GZ_NO_TRAILER:
			gz_set_fatal(st, NULL);	/* ARCHIVE_FATAL without a message (gzip.c:419-421) */
			cutoff = r->out_len == 0 ? (total / OUT_BLOCK) * OUT_BLOCK
			    : ((total + r->out_len - 1) / OUT_BLOCK) * OUT_BLOCK;
			last_out = r->out_len;
			take = i + 1;
			stop = 1;
			break;
		default:
			gz_set_fatal(st, "gzip decompression failed");
			cutoff = total;
			stop = 1;
			break;
		}
	}
	if (!stop) {
		/* every member of the window is fine: what comes after it? */
		if (x->end_kind == LA_END_EOF)
			st->eof = 1;
		else if (x->end_kind == LA_END_TRUNCATED) {
			gz_set_fatal(st, "truncated gzip input");
			cutoff = (total / OUT_BLOCK) * OUT_BLOCK;
		} else if (x->end_kind == LA_END_GZ_TOO_LARGE) {
			gz_set_fatal(st, la_end_message(LA_END_GZ_TOO_LARGE, 1));
			cutoff = (total / OUT_BLOCK) * OUT_BLOCK;
		}
	}

	/* ---- bring the bytes of members [0, take) behind the carry ---- */
	uint64_t new_bytes = (total - st->total_out) + last_out;
	const int ahead = st->ahead_ok && take && contiguous && last_out == 0 && st->carry_len == st->ahead_rem &&
	    new_bytes <= st->ahead_len;
	st->ahead_ok = 0;
	if (ahead) {
		/* they came over while the caller was busy (the sync above covered the copy): carry in front, change slabs */
		if (st->carry_len)
			memcpy(st->slab2, st->slab, st->carry_len);
		uint8_t *tp = st->slab; st->slab = st->slab2; st->slab2 = tp;
		size_t tc = st->slab_cap; st->slab_cap = st->slab2_cap; st->slab2_cap = tc;
	} else if (gz_grow_pinned(st, &st->slab, &st->slab_cap, st->carry_len + (size_t)new_bytes + 16, st->carry_len) < 0)
		return gz_gpu_fail(self, st, "pinned slab allocation");
	uint8_t *dstp = st->slab + st->carry_len;
	const double b2 = st->trace ? gz_now() : 0;
	if (take && !ahead) {
		if (contiguous && last_out == 0) {
			if (la_gpu_memcpy_d2h(st->gpu, dstp, st->d_dst, (size_t)new_bytes) != LA_OK)
				return gz_gpu_fail(self, st, "device to host copy");
		} else {
			size_t w = 0;
			for (uint32_t i = 0; i < take; i++) {
				size_t len = st->h_res[i].out_len;
				if (len && la_gpu_memcpy_d2h(st->gpu, dstp + w, (uint8_t *)st->d_dst + x->members[i].dst_off, len) != LA_OK)
					return gz_gpu_fail(self, st, "device to host copy");
				w += len;
			}
		}
		if (la_gpu_sync(st->gpu) != LA_OK)
			return gz_gpu_fail(self, st, "device to host copy");
	}
	if (st->trace)
		fprintf(stderr, "la_gzip:   h2d+decode %.1f ms, walk+grow %.1f ms, d2h %.1f ms (%llu bytes, contiguous %d, copied ahead %d)\n",
		    b1 - b0, b2 - b1, gz_now() - b2, (unsigned long long)new_bytes, contiguous, ahead);
	st->total_out = total + last_out;
	st->carry_len += (size_t)new_bytes;

	/* how much of [carry | new bytes] may go out now */
	uint64_t slab_start = st->total_out - st->carry_len;	/* stream offset of slab[0] */
	uint64_t lim;
	if (cutoff != UINT64_MAX)
		lim = cutoff;					/* an error follows: the reference's count */
	else if (st->eof)
		lim = st->total_out;				/* clean end: everything */
	else
		lim = (st->total_out / OUT_BLOCK) * OUT_BLOCK;	/* keep the partial last block back */
	if (lim < slab_start)
		lim = slab_start;
	st->last_ret = (size_t)(lim - slab_start);
	return 0;
}

/*
 * Gather one window, start its upload, find the member boundaries and queue the decode: nothing is
 * waited for.  Returns 1 when a window is in flight (st->idx, st->inflight), 0 when the state changed
 * instead (end of stream, pending error, wider window, looser boundary search: the caller looks again),
 * ARCHIVE_FATAL on error.
 */
static int gz_prepare(struct archive_read_filter *self, struct gzip_private *st)
{
	const double t0 = st->trace ? gz_now() : 0;
	while (!st->upstream_eof && st->stage_len < st->batch_bytes) {
		ssize_t avail;
		const void *up = __archive_read_filter_ahead(self->upstream, 1, &avail);
		if (up == NULL) {
			if (avail < 0)
				return ARCHIVE_FATAL;
			st->upstream_eof = 1;
			break;
		}
		size_t n = (size_t)avail;
		if (n > st->batch_bytes - st->stage_len)
			n = st->batch_bytes - st->stage_len;
		/* a stream that has already filled 8 MiB gets the whole window at once instead of
		 * five more rounds of pin-a-bigger-buffer-and-copy */
		size_t want = st->stage_len + n;
		if (want > ((size_t)8 << 20) && want < st->batch_bytes)
			want = st->batch_bytes;
		if (gz_grow_pinned(st, &st->stage, &st->stage_cap, want, st->stage_len) < 0)
			return gz_gpu_fail(self, st, "pinned staging allocation");
		memcpy(st->stage + st->stage_len, up, n);
		st->stage_len += n;
		__archive_read_filter_consume(self->upstream, (int64_t)n);
	}
	const double t1 = st->trace ? gz_now() : 0;
	/* the window goes to the device while the host looks for the member boundaries in it
	 * (stream-ordered copy from the pinned window; nothing writes to [0, stage_len)
	 * before the batch has been waited for) */
	if (st->stage_len &&
	    (gz_grow_dev(st, &st->d_src, &st->d_src_cap, st->stage_len + 64) < 0 ||
	     la_gpu_memcpy_h2d(st->gpu, st->d_src, st->stage, st->stage_len) != LA_OK))
		return gz_gpu_fail(self, st, "host to device copy");
	if (la_gz_index_build4(st->stage, st->stage_len, st->upstream_eof, st->hint_skip, st->hint_cap,
	    st->loose ? 0 : LA_GZ_INDEX_STRICT, st->out_budget, &st->idx) != 0) {
		archive_set_error(&self->archive->archive, ENOMEM, "Can't allocate data for gzip decompression");
		return ARCHIVE_FATAL;
	}
	if (st->idx.n == 0) {
		int kind = st->idx.end_kind;
		la_gz_index_free(&st->idx);
		if (la_gpu_sync(st->gpu) != LA_OK)	/* the upload above: the window may move now */
			return gz_gpu_fail(self, st, "host to device copy");
		if (kind == LA_END_NEED_MORE) {
			if (st->upstream_eof) { st->eof = 1; return 0; }
			if (!st->loose) {
				/* no trusted boundary in the whole window: before widening it, look
				 * with every 1f 8b 08 as a candidate (and keep doing so) */
				st->loose = 1;
				return 0;
			}
			st->batch_bytes *= 2;	/* one member larger than the window */
			return 0;
		}
		if (kind == LA_END_TRUNCATED)
			gz_set_fatal(st, "truncated gzip input");
		else if (kind == LA_END_GZ_TOO_LARGE)
			gz_set_fatal(st, la_end_message(LA_END_GZ_TOO_LARGE, 1));
		else
			st->eof = 1;
		return 0;
	}
	size_t used = 0;
	int rc = gzip_run_batch(self, st, &st->idx, &used, 0);
	if (rc < 0) {
		la_gz_index_free(&st->idx);
		return rc;
	}
	st->inflight = 1;
	/* decoded bytes of this window towards the OTHER slab, behind the place of what will be left of the carry: only for a
	 * window whose boundaries and sizes are the index's own (no guessed boundary, no retry hints) and of ordinary size */
	st->ahead_ok = 0;
	/* (not before the window ramp has reached its target: a second pinned slab costs about half a millisecond per MiB, and one
	 * that has to grow three times costs a short stream more than the copies it hides) */
	if (!st->no_ahead && st->batch_bytes >= st->target_bytes && !st->idx.speculative && st->hint_skip == 0 && st->hint_cap == 0 && st->idx.max_out != 0 &&
	    st->idx.max_out <= ((uint64_t)1 << 30) && st->carry_len >= st->last_ret) {
		const size_t rem = st->carry_len - st->last_ret;
		if (gz_grow_pinned(st, &st->slab2, &st->slab2_cap, rem + (size_t)st->idx.max_out + 16, 0) == 0 &&
		    la_gpu_memcpy_d2h(st->gpu, st->slab2 + rem, st->d_dst, (size_t)st->idx.max_out) == LA_OK) {
			st->ahead_ok = 1;
			st->ahead_rem = rem;
			st->ahead_len = (size_t)st->idx.max_out;
		}
	}
	if (st->batch_bytes < st->target_bytes)
		st->batch_bytes = st->batch_bytes * 2 < st->target_bytes ? st->batch_bytes * 2 : st->target_bytes;
	if (st->trace)
		fprintf(stderr, "la_gzip: window %zu bytes, %u members queued: gather %.1f ms, index + launch %.1f ms\n",
		    st->stage_len, st->idx.n, t1 - t0, gz_now() - t1);
	return 1;
}

static ssize_t gzip_filter_read(struct archive_read_filter *self, const void **p)
{
	struct gzip_private *st = (struct gzip_private *)self->data;
	*p = NULL;

	/* the bytes handed out last time are released now: close the gap */
	if (st->last_ret) {
		memmove(st->slab, st->slab + st->last_ret, st->carry_len - st->last_ret);
		st->carry_len -= st->last_ret;
		st->last_ret = 0;
	}
	for (;;) {
		if (st->pending_fatal) {
			if (st->pending_has_msg)
				archive_set_error(&self->archive->archive, ARCHIVE_ERRNO_MISC, "%s", st->pending_msg);
			return ARCHIVE_FATAL;
		}
		if (st->eof) {
			if (st->carry_len) {	/* the held-back tail of a clean stream */
				st->last_ret = st->carry_len;
				*p = st->slab;
				return (ssize_t)st->carry_len;
			}
			return 0;
		}
		if (!st->inflight) {
			if (st->upstream_failed)
				return ARCHIVE_FATAL;	/* upstream set the error when the window was gathered ahead */
			int pr = gz_prepare(self, st);
			if (pr < 0)
				return pr;
			if (pr == 0)
				continue;
		}
		/* the window in flight: results, stream-order walk, slab */
		size_t used = 0;
		const double t2 = st->trace ? gz_now() : 0;
		int rc = gzip_run_batch(self, st, &st->idx, &used, 1);
		st->inflight = 0;
		int made_progress = used > 0;
		if (st->trace)
			fprintf(stderr, "la_gzip:   finished in %.1f ms, used %zu of %zu, out %zu\n", gz_now() - t2, used, st->stage_len, st->last_ret);
		la_gz_index_free(&st->idx);
		if (rc < 0)
			return rc;
		if (used < st->stage_len)
			memmove(st->stage, st->stage + used, st->stage_len - used);
		st->stage_len -= used;
		if (!made_progress && !st->pending_fatal && !st->eof && st->hint_skip == 0 && st->hint_cap == 0) {
			/* nothing could be finished in this window: it has to grow */
			if (st->upstream_eof) { st->eof = 1; continue; }
			st->batch_bytes *= 2;
		}
		if (st->last_ret) {
			/* Decode ahead: gather, upload, index and launch the NEXT window before handing this
			 * slab out, so that the device works while the caller consumes it (the slab is not
			 * touched until the next read()).  An outcome other than "in flight" is simply met
			 * again by the next read(). */
			if (!st->pending_fatal && !st->eof) {
				int pr = gz_prepare(self, st);
				if (pr < 0)
					st->upstream_failed = 1;	/* reported by the next read(), after these bytes */
			}
			*p = st->slab;
			return (ssize_t)st->last_ret;
		}
	}
}

static int gzip_filter_close(struct archive_read_filter *self)
{
	struct gzip_private *st = (struct gzip_private *)self->data;
	if (st == NULL)
		return ARCHIVE_OK;
	if (st->gpu) {
		la_gpu_sync(st->gpu);
		if (st->inflight)
			la_gz_index_free(&st->idx);
		if (st->stage) la_gpu_free_host(st->gpu, st->stage);
		if (st->slab) la_gpu_free_host(st->gpu, st->slab);
		if (st->slab2) la_gpu_free_host(st->gpu, st->slab2);
		if (st->d_src) la_gpu_free(st->gpu, st->d_src);
		if (st->d_dst) la_gpu_free(st->gpu, st->d_dst);
		if (st->d_tabs) la_gpu_free(st->gpu, st->d_tabs);
		la_gpu_close(st->gpu);
	}
	free(st->h_res);
	free(st->name);
	free(st);
	self->data = NULL;
	return ARCHIVE_OK;
}
