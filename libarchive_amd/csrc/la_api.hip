/*
 * la_api.hip -- the extern "C" shim declared in include/la_gpu.h.
 *
 * Owns the HIP context objects (stream, events, workspace) and sequences the
 * kernels of one batch; all arithmetic lives in the kernel files.  No host
 * fallback exists: without a usable gfx950 device la_gpu_open() fails and every
 * caller above it fails loudly.
 */
#include "la_dev.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#define LA_MAX_SLICES 8
#define LA_GZ_LANES_MIN 8192u	/* members per batch from which the lane-per-member kernel is used (wave-per-member below: 4096 members 12.7 vs 23.8 ms, 16384 members 47.7 vs 26.0 ms) */
#define LA_PROF_MAX_RANGES 64

struct la_gpu_ctx {
	int device;
	hipStream_t own_stream;
	hipStream_t stream;
	hipEvent_t ev0, ev1;
	hipEvent_t mark;
	void *ws;
	uint64_t ws_bytes;
	/* second stream + events: block checksums / parse of later slices run beside the
	 * expand kernels of earlier ones */
	hipStream_t aux_stream;
	hipEvent_t slice_ev[LA_MAX_SLICES + 1];
	/* optional timing of the last batch: one (start, stop) event pair per kernel range,
	 * recorded on the stream the kernels run on, summed by name when read */
	int prof_on;
	int prof_n;
	hipEvent_t prof_a[LA_PROF_MAX_RANGES], prof_b[LA_PROF_MAX_RANGES];
	const char *prof_name[LA_PROF_MAX_RANGES];
	char err[256];
};

static void prof_begin(la_gpu_ctx *c) { c->prof_n = 0; }
/* returns a handle to close with prof_close, or -1 */
static int prof_open(la_gpu_ctx *c, const char *name, hipStream_t s)
{
	if (!c->prof_on || c->prof_n >= LA_PROF_MAX_RANGES)
		return -1;
	int h = c->prof_n++;
	c->prof_name[h] = name;
	(void)hipEventRecord(c->prof_a[h], s);
	return h;
}
static void prof_close(la_gpu_ctx *c, int h, hipStream_t s)
{
	if (h >= 0)
		(void)hipEventRecord(c->prof_b[h], s);
}

#define HIPCHK(ctx, call)                                                              \
	do {                                                                           \
		hipError_t e_ = (call);                                                \
		if (e_ != hipSuccess) {                                                \
			snprintf((ctx)->err, sizeof((ctx)->err), "%s: %s", #call,      \
			    hipGetErrorString(e_));                                    \
			return LA_ERR_HIP;                                             \
		}                                                                      \
	} while (0)

extern "C" {

int la_gpu_abi_version(void) { return LA_GPU_ABI_VERSION; }

int la_gpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int la_gpu_open(int device, la_gpu_ctx **out)
{
	if (!out)
		return LA_ERR_ARG;
	*out = NULL;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
		return LA_ERR_NO_DEVICE;
	la_gpu_ctx *c = new (std::nothrow) la_gpu_ctx();
	if (!c)
		return LA_ERR_NOMEM;
	memset(c, 0, sizeof(*c));
	c->device = device;
	if (hipSetDevice(device) != hipSuccess ||
	    hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
	    hipEventCreateWithFlags(&c->mark, hipEventDisableTiming) != hipSuccess) {
		delete c;
		return LA_ERR_NO_DEVICE;
	}
	for (int i = 0; i < LA_PROF_MAX_RANGES; i++) {
		(void)hipEventCreate(&c->prof_a[i]);
		(void)hipEventCreate(&c->prof_b[i]);
	}
	(void)hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
	for (int i = 0; i <= LA_MAX_SLICES; i++)
		(void)hipEventCreateWithFlags(&c->slice_ev[i], hipEventDisableTiming);
	c->stream = c->own_stream;
	*out = c;
	return LA_OK;
}

void la_gpu_close(la_gpu_ctx *c)
{
	if (!c)
		return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	if (c->ws) (void)hipFree(c->ws);
	(void)hipEventDestroy(c->ev0);
	(void)hipEventDestroy(c->ev1);
	(void)hipEventDestroy(c->mark);
	for (int i = 0; i < LA_PROF_MAX_RANGES; i++) {
		(void)hipEventDestroy(c->prof_a[i]);
		(void)hipEventDestroy(c->prof_b[i]);
	}
	for (int i = 0; i <= LA_MAX_SLICES; i++)
		(void)hipEventDestroy(c->slice_ev[i]);
	(void)hipStreamSynchronize(c->aux_stream);
	(void)hipStreamDestroy(c->aux_stream);
	(void)hipStreamDestroy(c->own_stream);
	delete c;
}

int la_gpu_set_stream(la_gpu_ctx *c, void *hip_stream)
{
	if (!c) return LA_ERR_ARG;
	c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
	return LA_OK;
}

int la_gpu_sync(la_gpu_ctx *c)
{
	if (!c) return LA_ERR_ARG;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return LA_OK;
}

const char *la_gpu_last_error(const la_gpu_ctx *c) { return c ? c->err : "no context"; }

int la_gpu_reserve(la_gpu_ctx *c, uint64_t bytes)
{
	if (!c) return LA_ERR_ARG;
	if (bytes <= c->ws_bytes)
		return LA_OK;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (c->ws) { HIPCHK(c, hipFree(c->ws)); c->ws = NULL; c->ws_bytes = 0; }
	bytes = (bytes + 0xFFFFFull) & ~0xFFFFFull;
	HIPCHK(c, hipMalloc(&c->ws, bytes));
	c->ws_bytes = bytes;
	return LA_OK;
}

int la_gpu_malloc(la_gpu_ctx *c, void **p, uint64_t bytes)
{
	if (!c || !p) return LA_ERR_ARG;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMalloc(p, bytes ? bytes : 1));
	return LA_OK;
}
int la_gpu_free(la_gpu_ctx *c, void *p)
{
	if (!c) return LA_ERR_ARG;
	if (p) HIPCHK(c, hipFree(p));
	return LA_OK;
}
int la_gpu_malloc_host(la_gpu_ctx *c, void **p, uint64_t bytes)
{
	if (!c || !p) return LA_ERR_ARG;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
	return LA_OK;
}
int la_gpu_free_host(la_gpu_ctx *c, void *p)
{
	if (!c) return LA_ERR_ARG;
	if (p) HIPCHK(c, hipHostFree(p));
	return LA_OK;
}
int la_gpu_memcpy_h2d(la_gpu_ctx *c, void *d, const void *h, uint64_t bytes)
{
	if (!c) return LA_ERR_ARG;
	if (bytes) HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
	return LA_OK;
}
int la_gpu_memcpy_d2h(la_gpu_ctx *c, void *h, const void *d, uint64_t bytes)
{
	if (!c) return LA_ERR_ARG;
	if (bytes) HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
	return LA_OK;
}

int la_gpu_mark(la_gpu_ctx *c)
{
	if (!c) return LA_ERR_ARG;
	HIPCHK(c, hipEventRecord(c->mark, c->stream));
	return LA_OK;
}
int la_gpu_wait_mark(la_gpu_ctx *c)
{
	if (!c) return LA_ERR_ARG;
	HIPCHK(c, hipEventSynchronize(c->mark));
	return LA_OK;
}

int la_gpu_memcpy_d2d(la_gpu_ctx *c, void *d, const void *src_, uint64_t bytes)
{
	if (!c) return LA_ERR_ARG;
	if (bytes) HIPCHK(c, hipMemcpyAsync(d, src_, bytes, hipMemcpyDeviceToDevice, c->stream));
	return LA_OK;
}

int la_gpu_timer_start(la_gpu_ctx *c)
{
	if (!c) return LA_ERR_ARG;
	HIPCHK(c, hipEventRecord(c->ev0, c->stream));
	return LA_OK;
}
int la_gpu_timer_stop(la_gpu_ctx *c, float *ms)
{
	if (!c || !ms) return LA_ERR_ARG;
	HIPCHK(c, hipEventRecord(c->ev1, c->stream));
	HIPCHK(c, hipEventSynchronize(c->ev1));
	HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
	return LA_OK;
}

int la_gpu_profile_enable(la_gpu_ctx *c, int on)
{
	if (!c) return LA_ERR_ARG;
	c->prof_on = on ? 1 : 0;
	c->prof_n = 0;
	return LA_OK;
}

int la_gpu_profile_read(la_gpu_ctx *c, float *ms, const char **names, int cap)
{
	if (!c || !ms) return LA_ERR_ARG;
	if (!c->prof_on || c->prof_n == 0)
		return 0;
	int n = 0;
	for (int i = 0; i < c->prof_n; i++) {
		float t = 0;
		HIPCHK(c, hipEventSynchronize(c->prof_b[i]));
		HIPCHK(c, hipEventElapsedTime(&t, c->prof_a[i], c->prof_b[i]));
		int k = 0;
		for (; k < n; k++)
			if (names && names[k] == c->prof_name[i])
				break;
		if (k == n) {
			if (n >= cap)
				continue;
			if (names) names[n] = c->prof_name[i];
			ms[n] = 0;
			n++;
		}
		ms[k] += t;
	}
	return n;
}

/* ------------------------------------------------------------------ hashes */

int la_gpu_xxh32_many(la_gpu_ctx *c, const uint8_t *d_base, const la_hash_job *d_jobs,
    uint32_t n, uint32_t *d_out)
{
	if (!c || (n && (!d_base || !d_jobs || !d_out))) return LA_ERR_ARG;
	la_launch_xxh32_many(c->stream, d_base, d_jobs, n, d_out);
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

int la_gpu_crc32_many(la_gpu_ctx *c, const uint8_t *d_base, const la_hash_job *d_jobs,
    uint32_t n, uint32_t *d_out)
{
	if (!c || (n && (!d_base || !d_jobs || !d_out))) return LA_ERR_ARG;
	la_launch_crc32_many(c->stream, d_base, d_jobs, n, d_out);
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

/* ------------------------------------------------------------------ lz4 */

static uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) & ~(a - 1); }

/* workspace layout of one lz4 batch */
struct lz4_ws {
	uint32_t *nseq;		/* [n] */
	uint32_t *caps;		/* [n] table capacity per block */
	uint32_t *sum_status;	/* [n] block-checksum verdicts (merged into the status at the end) */
	uint32_t *big;		/* [n+1] count + list of blocks with more sequences than one LDS segment holds */
	uint64_t *table_off;	/* [n+1] */
	void *scan;		/* scan scratch */
	la_lz4_seq *table;
	uint64_t table_cap;	/* entries */
	uint64_t total;
};

static void lz4_ws_layout(lz4_ws *w, uint8_t *base, uint32_t n, uint64_t src_bytes, bool with_table)
{
	uint64_t o = 0;
	w->nseq = (uint32_t *)(base + o); o += align_up((uint64_t)n * 4, 256);
	w->caps = (uint32_t *)(base + o); o += align_up((uint64_t)n * 4, 256);
	w->sum_status = (uint32_t *)(base + o); o += align_up((uint64_t)n * 4, 256);
	w->big = (uint32_t *)(base + o); o += align_up((uint64_t)(n + 1) * 4, 256);
	w->table_off = (uint64_t *)(base + o); o += align_up(((uint64_t)n + 1) * 8, 256);
	w->scan = base + o; o += align_up(la_scan_scratch_bytes(n), 256);
	/* a non-final sequence takes >= 3 payload bytes; slots are rounded up to 8 entries:
	 * sum((src_len/3 + 1 + 7) & ~7) <= src_bytes/3 + 8n */
	w->table_cap = with_table ? src_bytes / 3 + 8ull * n : 0;
	w->table = (la_lz4_seq *)(base + o); o += align_up(w->table_cap * sizeof(la_lz4_seq), 256);
	w->total = o + 4096;
}

uint64_t la_gpu_lz4_workspace_bytes(uint32_t n_blocks, uint64_t src_bytes)
{
	lz4_ws w;
	lz4_ws_layout(&w, NULL, n_blocks, src_bytes, true);
	return w.total;
}

int la_gpu_lz4_decode(la_gpu_ctx *c, const la_lz4_batch *bt)
{
	if (!c || !bt)
		return LA_ERR_ARG;
	if (bt->n_blocks && (!bt->d_src || !bt->d_blocks || !bt->d_dst || !bt->d_out_len ||
	    !bt->d_dst_off || !bt->d_block_status))
		return LA_ERR_ARG;
	if (bt->n_frames && (!bt->d_frames || !bt->d_frame_status))
		return LA_ERR_ARG;
	if (!bt->d_dst_off)
		return LA_ERR_ARG;
	const bool fast = !(bt->options & LA_LZ4_OPT_GENERAL_ONLY);
	const bool verify = !(bt->options & LA_LZ4_OPT_NO_VERIFY);
	/* the in-order expand kernel runs on request (cross-check); not for images of less than 16 bytes */
	const bool poll = (bt->options & LA_LZ4_OPT_EXPAND_INORDER) == 0 || !la_lz4_expand_inorder_takes(bt->src_bytes);
	lz4_ws w;
	lz4_ws_layout(&w, NULL, bt->n_blocks, bt->src_bytes, fast);
	if (w.total > c->ws_bytes) {
		int rc = la_gpu_reserve(c, w.total);
		if (rc != LA_OK) return rc;
	}
	lz4_ws_layout(&w, (uint8_t *)c->ws, bt->n_blocks, bt->src_bytes, fast);
	hipStream_t sx = c->stream;		/* main stream (the caller's): checksums, expand, summary */
	hipStream_t sp = c->aux_stream;		/* second stream: parse beside the block checksums, frame
						 * checksums beside the expand kernel */
	const uint32_t n = bt->n_blocks;
	int h;

	prof_begin(c);
	/* the second stream starts after whatever the caller queued on its stream */
	HIPCHK(c, hipEventRecord(c->slice_ev[LA_MAX_SLICES], sx));
	HIPCHK(c, hipStreamWaitEvent(sp, c->slice_ev[LA_MAX_SLICES], 0));

	/* second stream: token-chain parse (+ sequence tables) of the whole batch.  One lane per
	 * block: it needs the whole table in one launch to fill the chip. */
	if (n)
		HIPCHK(c, hipMemsetAsync(bt->d_block_status, 0, (size_t)n * sizeof(uint32_t), sp));
	if (fast) {
		la_launch_lz4_table_caps(sp, bt->d_blocks, n, w.caps);
		la_launch_scan_u32(sp, w.caps, n, w.table_off, w.scan);
	}
	/* block checksums (into their own verdict array) and the token-chain parse: one fused
	 * kernel that stages the image through LDS and reads it from HBM once; the
	 * first-generation pair of kernels stays selectable as a cross-check */
	if (verify && n)
		HIPCHK(c, hipMemsetAsync(w.sum_status, 0, (size_t)n * sizeof(uint32_t), sp));
	if (bt->options & LA_LZ4_OPT_PARSE_V1) {
		if (verify && n) {
			h = prof_open(c, "lz4_block_sums", sp);
			la_launch_lz4_block_sums(sp, bt->d_src, bt->d_blocks, n, w.sum_status);
			prof_close(c, h, sp);
		}
		h = prof_open(c, "lz4_parse", sp);
		la_launch_lz4_parse(sp, bt->d_src, bt->src_bytes, bt->d_blocks, n, bt->d_out_len, w.nseq,
		    bt->d_block_status, fast ? w.table : NULL, w.table_off, w.table_cap);
		prof_close(c, h, sp);
	} else {
		h = prof_open(c, "lz4_parse", sp);
		la_launch_lz4_parse_staged(sp, bt->d_src, bt->src_bytes, bt->d_blocks, n, bt->d_out_len, w.nseq,
		    bt->d_block_status, verify ? w.sum_status : NULL, fast ? w.table : NULL, w.table_off, w.table_cap);
		prof_close(c, h, sp);
	}
	HIPCHK(c, hipEventRecord(c->slice_ev[0], sp));

	HIPCHK(c, hipStreamWaitEvent(sx, c->slice_ev[0], 0));
	h = prof_open(c, "scan", sx);
	la_launch_scan_u32(sx, bt->d_out_len, n, bt->d_dst_off, w.scan);
	prof_close(c, h, sx);

	/* blocks the LDS-window kernel does not take (any size, stored, chains of dependent
	 * blocks) first, over the whole table; then the LDS-window kernel in slices, each
	 * slice's frames hashed on the second stream while the next slice expands */
	h = prof_open(c, fast ? "lz4_expand_general" : "lz4_expand", sx);
	la_launch_lz4_expand_general(sx, bt->d_src, bt->src_bytes, bt->d_blocks, n, bt->d_dst,
	    bt->dst_cap, bt->d_dst_off, bt->d_out_len, bt->d_block_status, w.nseq,
	    fast ? 0xFFFFFFFEu : 0u,	/* the LDS-window kernel takes every eligible block that got a table ... */
	    bt->hist_len, LA_LZ4_LONG_SEQ_BYTES);	/* ... except blocks of few long sequences (la_dev.h) */
	prof_close(c, h, sx);
	if (fast && poll) {
		/* eligible blocks with more sequences than one LDS segment: classified on the device, shared out over a
		 * small grid (a no-op launch when there are none).  (The in-order kernel takes blocks of any sequence count.) */
		h = prof_open(c, "lz4_expand_big", sx);
		la_launch_lz4_expand_fast_big(sx, bt->d_src, bt->src_bytes, bt->d_blocks, n, bt->d_dst, bt->dst_cap,
		    bt->d_dst_off, bt->d_out_len, bt->d_block_status, w.nseq, w.table, w.table_off, w.big, LA_LZ4_LONG_SEQ_BYTES);
		prof_close(c, h, sx);
	}
	const uint32_t nsl = (fast && n >= 4u * 8192u) ? 4u : 1u;
	if (bt->n_frames && !verify)
		HIPCHK(c, hipMemsetAsync(bt->d_frame_status, 0, (size_t)bt->n_frames * sizeof(uint32_t), sx));
	for (uint32_t i = 0; i < nsl; i++) {
		const uint32_t first = (uint32_t)((uint64_t)n * i / nsl), last = (uint32_t)((uint64_t)n * (i + 1) / nsl);
		if (fast) {
			h = prof_open(c, "lz4_expand", sx);
			(poll ? la_launch_lz4_expand_fast : la_launch_lz4_expand_inorder)(sx, bt->d_src, bt->src_bytes, bt->d_blocks + first, last - first, bt->d_dst,
			    bt->dst_cap, bt->d_dst_off + first, bt->d_out_len + first, bt->d_block_status + first,
			    w.nseq + first, w.table, w.table_off + first, LA_LZ4_LONG_SEQ_BYTES);
			prof_close(c, h, sx);
		}
		if (bt->n_frames && verify) {
			HIPCHK(c, hipEventRecord(c->slice_ev[1 + i], sx));
			HIPCHK(c, hipStreamWaitEvent(sp, c->slice_ev[1 + i], 0));
			h = prof_open(c, "lz4_frame_sums", sp);
			/* frames that END in this slice: all their blocks are in the slab now */
			la_launch_lz4_frame_sums(sp, bt->d_src, bt->d_dst, bt->d_frames, bt->n_frames,
			    bt->d_dst_off, bt->dst_cap, bt->d_frame_status, i ? first + 1 : 0u, last,
			    bt->d_carry_in, bt->d_carry_out,
			    /* nothing overlaps the last slice's hashes (nor a small batch's): the low-latency form */
			    i + 1 == nsl);
			prof_close(c, h, sp);
		}
	}
	/* join the second stream */
	HIPCHK(c, hipEventRecord(c->slice_ev[LA_MAX_SLICES - 1], sp));
	HIPCHK(c, hipStreamWaitEvent(sx, c->slice_ev[LA_MAX_SLICES - 1], 0));
	if (verify)
		la_launch_lz4_merge_status(sx, w.sum_status, n, bt->d_block_status);
	if (bt->d_summary) {
		h = prof_open(c, "summary", sx);
		la_launch_lz4_summary(sx, bt->d_out_len, bt->d_block_status, n,
		    bt->d_frame_status, bt->n_frames, bt->d_dst_off, bt->d_summary);
		prof_close(c, h, sx);
	}
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

uint64_t la_gpu_zstd_workspace_bytes(uint32_t n_frames)
{
	return la_zstd_workspace_bytes(n_frames);
}

int la_gpu_zstd_decode(la_gpu_ctx *c, const la_zstd_batch *bt)
{
	if (!c || !bt)
		return LA_ERR_ARG;
	if (bt->n_frames && (!bt->d_src || !bt->d_frames || !bt->d_dst || !bt->d_results))
		return LA_ERR_ARG;
	const uint64_t need = la_zstd_workspace_bytes(bt->n_frames);
	if (need > c->ws_bytes) {
		int rc = la_gpu_reserve(c, need);
		if (rc != LA_OK) return rc;
	}
	prof_begin(c);
	int h = prof_open(c, "zstd_frames", c->stream);
	la_launch_zstd_frames(c->stream, bt->d_src, bt->src_bytes, bt->d_frames, bt->n_frames, bt->d_dst, bt->dst_cap,
	    bt->d_results, (uint8_t *)c->ws, bt->options);
	prof_close(c, h, c->stream);
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

int la_gpu_gzip_decode(la_gpu_ctx *c, const la_gz_batch *bt)
{
	if (!c || !bt)
		return LA_ERR_ARG;
	if (bt->n_members && (!bt->d_src || !bt->d_members || !bt->d_dst || !bt->d_results))
		return LA_ERR_ARG;
	hipStream_t s = c->stream;
	/* many members: one LANE per member (la_inflate_lanes.hip); few: one wave per member */
	const bool lanes = (bt->n_members >= LA_GZ_LANES_MIN || (bt->options & (LA_GZ_OPT_LANE_KERNEL | LA_GZ_OPT_TWO_PHASE))) &&
	    !(bt->options & LA_GZ_OPT_WAVE_KERNEL);
	/* lanes, two phases (default): entropy decode into literal buffers + sequence tables, then the
	 * LDS-window expand kernel of the lz4 path; LA_GZ_OPT_LANE_KERNEL forces the in-place lane kernel */
	const bool two_phase = lanes && !(bt->options & LA_GZ_OPT_LANE_KERNEL);
	const uint32_t n = bt->n_members;
	uint8_t *wsb = NULL;
	la_inflate_emit E = {};
	uint32_t *gz_big = NULL;
	if (lanes) {
		uint64_t need = align_up(la_inflate_lanes_scratch_bytes(n), 256);
		const uint64_t o_scratch = 0;
		uint64_t o = need;
		uint64_t o_lit = 0, o_tab = 0, o_blk = 0, o_olen = 0, o_nseq = 0, o_xst = 0, o_todo = 0, o_doff = 0, o_toff = 0, o_big = 0;
		if (two_phase) {
			o_lit = o;  o += align_up((uint64_t)n * 65536u, 256);
			o_tab = o;  o += align_up((uint64_t)n * LA_INFLATE_MAXSEQ * sizeof(la_lz4_seq), 256);
			o_blk = o;  o += align_up((uint64_t)n * sizeof(la_lz4_block), 256);
			o_olen = o; o += align_up((uint64_t)n * 4, 256);
			o_nseq = o; o += align_up((uint64_t)n * 4, 256);
			o_xst = o;  o += align_up((uint64_t)n * 4, 256);
			o_todo = o; o += align_up((uint64_t)n * 4, 256);
			o_doff = o; o += align_up((uint64_t)(n + 1) * 8, 256);
			o_toff = o; o += align_up((uint64_t)(n + 1) * 8, 256);
			o_big = o;  o += align_up((uint64_t)(n + 1) * 4, 256);
		}
		if (o > c->ws_bytes) {
			int rc = la_gpu_reserve(c, o);
			if (rc != LA_OK) return rc;
		}
		wsb = (uint8_t *)c->ws;
		(void)o_scratch;
		if (two_phase) {
			E.lit = wsb + o_lit;
			E.table = (la_lz4_seq *)(wsb + o_tab);
			E.blocks = (la_lz4_block *)(wsb + o_blk);
			E.out_len = (uint32_t *)(wsb + o_olen);
			E.nseq = (uint32_t *)(wsb + o_nseq);
			E.xstatus = (uint32_t *)(wsb + o_xst);
			E.todo = (uint32_t *)(wsb + o_todo);
			E.dst_off = (uint64_t *)(wsb + o_doff);
			E.table_off = (uint64_t *)(wsb + o_toff);
			gz_big = (uint32_t *)(wsb + o_big);
		}
	}
	prof_begin(c);
	int h;
	if (two_phase) {
		h = prof_open(c, "inflate_symbols", s);
		la_launch_inflate_symbols(s, bt->d_src, bt->src_bytes, bt->d_members, n, bt->dst_cap, bt->d_results, wsb, E);
		prof_close(c, h, s);
		h = prof_open(c, "inflate_expand", s);
		if (!(bt->options & LA_GZ_OPT_EXPAND_INORDER)) {
			la_launch_lz4_expand_fast(s, E.lit, (uint64_t)n * 65536u, E.blocks, n, bt->d_dst, bt->dst_cap,
			    E.dst_off, E.out_len, E.xstatus, E.nseq, E.table, E.table_off, 0u);
			/* members with more matches than one LDS segment of the polling kernel holds */
			la_launch_lz4_expand_fast_big(s, E.lit, (uint64_t)n * 65536u, E.blocks, n, bt->d_dst, bt->dst_cap,
			    E.dst_off, E.out_len, E.xstatus, E.nseq, E.table, E.table_off, gz_big, 0u);
		} else {
			la_launch_lz4_expand_inorder(s, E.lit, (uint64_t)n * 65536u, E.blocks, n, bt->d_dst, bt->dst_cap,
			    E.dst_off, E.out_len, E.xstatus, E.nseq, E.table, E.table_off, 0u);
		}
		prof_close(c, h, s);
		/* members the LDS-window kernel cannot take: decoded in place */
		h = prof_open(c, "inflate", s);
		la_launch_inflate_lanes(s, bt->d_src, bt->src_bytes, bt->d_members, n, bt->d_dst,
		    bt->dst_cap, bt->d_results, wsb, E.todo);
		prof_close(c, h, s);
	} else {
		h = prof_open(c, "inflate", s);
		if (lanes)
			la_launch_inflate_lanes(s, bt->d_src, bt->src_bytes, bt->d_members, n, bt->d_dst,
			    bt->dst_cap, bt->d_results, wsb, NULL);
		else
			la_launch_inflate(s, bt->d_src, bt->src_bytes, bt->d_members, n, bt->d_dst, bt->dst_cap,
			    bt->d_results);
		prof_close(c, h, s);
	}
	h = prof_open(c, "gz_crc32", s);
	la_launch_gz_verify(s, bt->d_src, bt->src_bytes, bt->d_members, bt->n_members, bt->d_dst,
	    bt->d_results, (bt->options & LA_GZ_OPT_RAW) ? 2 : !(bt->options & LA_GZ_OPT_NO_VERIFY));
	prof_close(c, h, s);
	if (bt->d_summary)
		la_launch_gz_summary(s, bt->d_results, bt->n_members, bt->d_summary);
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

/* ------------------------------------------------------------------ lz4 compression */

uint64_t la_gpu_lz4_compress_bound(uint64_t src_bytes, uint32_t block_size, uint32_t bpf)
{
	if (block_size == 0 || bpf == 0)
		return 0;
	const uint64_t nb = (src_bytes + block_size - 1) / block_size, nf = (nb + bpf - 1) / bpf;
	return src_bytes + nb * (block_size / 255u + 16u + 8u) + nf * 15u + 64u;
}

int la_gpu_lz4_compress(la_gpu_ctx *c, const la_lz4c_batch *bt)
{
	if (!c || !bt || !bt->d_out_bytes || (bt->src_bytes && (!bt->d_src || !bt->d_out)))
		return LA_ERR_ARG;
	if (bt->block_size == 0 || bt->block_size > 65536u || bt->blocks_per_frame == 0 ||
	    (uint64_t)bt->block_size * bt->blocks_per_frame > 0x7FFFFFFFull ||
	    (bt->src_bytes + bt->block_size - 1) / bt->block_size > 0xFFFFFFFEull)
		return LA_ERR_ARG;
	const uint64_t need = la_gpu_lz4_compress_workspace_bytes(bt->src_bytes, bt->block_size, bt->blocks_per_frame);
	if (need > c->ws_bytes) {
		int rc = la_gpu_reserve(c, need);
		if (rc != LA_OK) return rc;
	}
	prof_begin(c);
	int h = prof_open(c, "lz4_compress", c->stream);
	la_launch_lz4_compress(c->stream, bt->d_src, bt->src_bytes, bt->block_size, bt->blocks_per_frame, bt->flags,
	    bt->d_out, bt->out_cap, bt->d_out_bytes, (uint8_t *)c->ws);
	prof_close(c, h, c->stream);
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

/* ------------------------------------------------------------------ gzip compression */

uint64_t la_gpu_gzip_compress_bound(uint64_t src_bytes, uint32_t chunk)
{
	if (chunk == 0)
		return 0;
	const uint64_t nc = (src_bytes + chunk - 1) / chunk;
	return src_bytes + nc * (18u + 8u + 5u) + 64u;	/* a chunk that does not shrink is stored: 5 bytes of block header */
}

int la_gpu_gzip_compress(la_gpu_ctx *c, const la_gzc_batch *bt)
{
	if (!c || !bt || !bt->d_out_bytes || (bt->src_bytes && (!bt->d_src || !bt->d_out)))
		return LA_ERR_ARG;
	if (bt->chunk_bytes == 0 || bt->chunk_bytes > 49152u || (bt->src_bytes + bt->chunk_bytes - 1) / bt->chunk_bytes > 0xFFFFFFFEull)
		return LA_ERR_ARG;
	const uint64_t need = la_gpu_gzip_compress_workspace_bytes(bt->src_bytes, bt->chunk_bytes);
	if (need > c->ws_bytes) {
		int rc = la_gpu_reserve(c, need);
		if (rc != LA_OK) return rc;
	}
	prof_begin(c);
	int h = prof_open(c, "gzip_compress", c->stream);
	la_launch_gzip_compress(c->stream, bt->d_src, bt->src_bytes, bt->chunk_bytes, bt->mtime, bt->d_out, bt->out_cap,
	    bt->d_out_bytes, (uint8_t *)c->ws);
	prof_close(c, h, c->stream);
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

} /* extern "C" */
