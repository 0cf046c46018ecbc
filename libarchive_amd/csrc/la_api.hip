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

struct la_gpu_ctx {
	int device;
	hipStream_t own_stream;
	hipStream_t stream;
	hipEvent_t ev0, ev1;
	void *ws;
	uint64_t ws_bytes;
	/* optional per-phase timing of the last batch (HIP events on the work stream) */
	int prof_on;
	int prof_n;
	hipEvent_t prof_ev[LA_PROF_MAX_PHASES + 1];
	const char *prof_name[LA_PROF_MAX_PHASES];
	char err[256];
};

static void prof_begin(la_gpu_ctx *c)
{
	c->prof_n = 0;
	if (c->prof_on)
		(void)hipEventRecord(c->prof_ev[0], c->stream);
}
static void prof_mark(la_gpu_ctx *c, const char *name)
{
	if (!c->prof_on || c->prof_n >= LA_PROF_MAX_PHASES)
		return;
	c->prof_name[c->prof_n] = name;
	c->prof_n++;
	(void)hipEventRecord(c->prof_ev[c->prof_n], c->stream);
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
	    hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
		delete c;
		return LA_ERR_NO_DEVICE;
	}
	for (int i = 0; i <= LA_PROF_MAX_PHASES; i++)
		(void)hipEventCreate(&c->prof_ev[i]);
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
	for (int i = 0; i <= LA_PROF_MAX_PHASES; i++)
		(void)hipEventDestroy(c->prof_ev[i]);
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
	HIPCHK(c, hipEventSynchronize(c->prof_ev[c->prof_n]));
	int n = c->prof_n < cap ? c->prof_n : cap;
	for (int i = 0; i < n; i++) {
		HIPCHK(c, hipEventElapsedTime(&ms[i], c->prof_ev[i], c->prof_ev[i + 1]));
		if (names) names[i] = c->prof_name[i];
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
	uint32_t *lcaps;	/* [n] literal-index capacity per block */
	uint64_t *table_off;	/* [n+1] */
	uint64_t *lidx_off;	/* [n+1] */
	void *scan;		/* scan scratch */
	la_lz4_seq *table;
	uint64_t table_cap;	/* entries */
	uint16_t *lidx;
	uint64_t lidx_cap;	/* entries */
	uint64_t total;
};

static void lz4_ws_layout(lz4_ws *w, uint8_t *base, uint32_t n, uint64_t src_bytes, bool with_table)
{
	uint64_t o = 0;
	w->nseq = (uint32_t *)(base + o); o += align_up((uint64_t)n * 4, 256);
	w->caps = (uint32_t *)(base + o); o += align_up((uint64_t)n * 4, 256);
	w->lcaps = (uint32_t *)(base + o); o += align_up((uint64_t)n * 4, 256);
	w->table_off = (uint64_t *)(base + o); o += align_up(((uint64_t)n + 1) * 8, 256);
	w->lidx_off = (uint64_t *)(base + o); o += align_up(((uint64_t)n + 1) * 8, 256);
	w->scan = base + o; o += align_up(la_scan_scratch_bytes(n), 256);
	/* a non-final sequence takes >= 3 payload bytes: sum(src_len/3 + 1) <= src_bytes/3 + n */
	w->table_cap = with_table ? src_bytes / 3 + n : 0;
	w->table = (la_lz4_seq *)(base + o); o += align_up(w->table_cap * sizeof(la_lz4_seq), 256);
	/* one u16 per 16 payload bytes: sum((src_len+15)/16) <= src_bytes/16 + n */
	w->lidx_cap = with_table ? src_bytes / 16 + n : 0;
	w->lidx = (uint16_t *)(base + o); o += align_up(w->lidx_cap * sizeof(uint16_t), 256);
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
	lz4_ws w;
	lz4_ws_layout(&w, NULL, bt->n_blocks, bt->src_bytes, fast);
	if (w.total > c->ws_bytes) {
		int rc = la_gpu_reserve(c, w.total);
		if (rc != LA_OK) return rc;
	}
	lz4_ws_layout(&w, (uint8_t *)c->ws, bt->n_blocks, bt->src_bytes, fast);
	hipStream_t s = c->stream;

	prof_begin(c);
	if (bt->n_blocks)
		HIPCHK(c, hipMemsetAsync(bt->d_block_status, 0, (size_t)bt->n_blocks * sizeof(uint32_t), s));
	if (verify) {
		la_launch_lz4_block_sums(s, bt->d_src, bt->d_blocks, bt->n_blocks, bt->d_block_status);
		prof_mark(c, "lz4_block_sums");
	}
	if (fast) {
		la_launch_lz4_table_caps(s, bt->d_blocks, bt->n_blocks, w.caps, w.lcaps);
		la_launch_scan_u32(s, w.caps, bt->n_blocks, w.table_off, w.scan);
		la_launch_scan_u32(s, w.lcaps, bt->n_blocks, w.lidx_off, w.scan);
	}
	la_launch_lz4_parse(s, bt->d_src, bt->src_bytes, bt->d_blocks, bt->n_blocks, bt->d_out_len, w.nseq,
	    bt->d_block_status, fast ? w.table : NULL, w.table_off, w.table_cap, w.lidx, w.lidx_off, w.lidx_cap);
	prof_mark(c, "lz4_parse");
	la_launch_scan_u32(s, bt->d_out_len, bt->n_blocks, bt->d_dst_off, w.scan);
	prof_mark(c, "scan");
	if (fast) {
		la_launch_lz4_expand_fast(s, bt->d_src, bt->src_bytes, bt->d_blocks, bt->n_blocks, bt->d_dst,
		    bt->dst_cap, bt->d_dst_off, bt->d_out_len, bt->d_block_status, w.nseq, w.table, w.table_off,
		    w.lidx, w.lidx_off);
		prof_mark(c, "lz4_expand");
	}
	la_launch_lz4_expand_general(s, bt->d_src, bt->src_bytes, bt->d_blocks, bt->n_blocks, bt->d_dst,
	    bt->dst_cap, bt->d_dst_off, bt->d_out_len, bt->d_block_status, w.nseq,
	    fast ? LA_LZ4_FAST_MAXSEQ : 0u);
	prof_mark(c, fast ? "lz4_expand_general" : "lz4_expand");
	if (bt->n_frames) {
		if (verify)
			la_launch_lz4_frame_sums(s, bt->d_src, bt->d_dst, bt->d_frames, bt->n_frames,
			    bt->d_dst_off, bt->dst_cap, bt->d_frame_status);
		else
			HIPCHK(c, hipMemsetAsync(bt->d_frame_status, 0, (size_t)bt->n_frames * sizeof(uint32_t), s));
	}
	prof_mark(c, "lz4_frame_sums");
	if (bt->d_summary)
		la_launch_lz4_summary(s, bt->d_out_len, bt->d_block_status, bt->n_blocks,
		    bt->d_frame_status, bt->n_frames, bt->d_dst_off, bt->d_summary);
	prof_mark(c, "summary");
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
	prof_begin(c);
	la_launch_inflate(s, bt->d_src, bt->src_bytes, bt->d_members, bt->n_members, bt->d_dst, bt->dst_cap,
	    bt->d_results);
	prof_mark(c, "inflate");
	la_launch_gz_verify(s, bt->d_src, bt->src_bytes, bt->d_members, bt->n_members, bt->d_dst,
	    bt->d_results, !(bt->options & LA_GZ_OPT_NO_VERIFY));
	prof_mark(c, "gz_crc32");
	if (bt->d_summary)
		la_launch_gz_summary(s, bt->d_results, bt->n_members, bt->d_summary);
	prof_mark(c, "summary");
	HIPCHK(c, hipGetLastError());
	return LA_OK;
}

} /* extern "C" */
