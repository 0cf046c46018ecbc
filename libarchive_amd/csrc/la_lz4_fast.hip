/*
 * la_lz4_fast.hip -- LDS-window LZ4 expand kernel (gfx950): the hot kernel of
 * the lz4 filter path for independent blocks of at most 64 KiB
 * (libarchive/archive_read_support_filter_lz4.c:557-561, the
 * LZ4_decompress_safe call of BD=4 frames).
 *
 * One 512-thread workgroup owns one block.  The parse kernel has already
 * reduced the token chain to a table of sequences {literal source, literal
 * length, output position, match offset}, so inside the block every copy is
 * known up front:
 *   - the block's whole 64 KiB output lives in an LDS window (two workgroups
 *     per CU: 2 x 76 KiB of the 160 KiB LDS);
 *   - one THREAD owns one sequence: it copies its literals from the
 *     compressed payload (HBM/L2, 8-byte loads) into the window, then its
 *     match inside the window;
 *   - a match may read bytes that an earlier match produces.  Instead of
 *     decoding in order, every sequence publishes a "done" byte in LDS and a
 *     match waits only for the (typically one to three) earlier sequences
 *     that overlap its source range, found by a binary search over the
 *     sequences' output positions.  Dependencies always point to lower
 *     sequence numbers and waves take sequences in increasing order, so the
 *     lowest unfinished sequence can always run: no deadlock, no barrier in
 *     the main loop;
 *   - finally the window is streamed to the decoded slab with 16-byte
 *     coalesced stores (window placed so that LDS and HBM addresses are
 *     congruent modulo 16).
 * HBM traffic per block: payload once + table once in, decoded bytes once out.
 */
#include "la_dev.h"

#define FAST_THREADS 512
#define FAST_WAVES   (FAST_THREADS / 64)

__device__ __forceinline__ uint64_t ld_u64(const uint8_t *p)
{
	uint64_t v;
	__builtin_memcpy(&v, p, 8);
	return v;
}

template <uint32_t MAXSEQ>
__global__ __launch_bounds__(FAST_THREADS) void lz4_expand_fast_kernel(
    const uint8_t *__restrict__ src, uint64_t src_bytes, const la_lz4_block *__restrict__ blocks,
    uint32_t n, uint8_t *__restrict__ dst, uint64_t dst_cap, const uint64_t *__restrict__ dst_off,
    const uint32_t *__restrict__ out_len, const uint32_t *__restrict__ status,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table,
    const uint64_t *__restrict__ table_off)
{
	__shared__ __attribute__((aligned(16))) uint8_t win[65536 + 16];
	__shared__ uint16_t dstpos[MAXSEQ + 4];
	__shared__ uint8_t done[MAXSEQ];

	const uint32_t bi = blockIdx.x;
	if (bi >= n)
		return;
	const la_lz4_block b = blocks[bi];
	const uint32_t olen = out_len[bi];
	const uint32_t ns = nseq[bi];
	const uint64_t doff = dst_off[bi];
	/* same predicate as the general kernel's skip test */
	if (status[bi] != LA_ST_OK || olen == 0 || !la_lz4_fast_eligible(b) || ns > MAXSEQ ||
	    doff + olen > dst_cap)
		return;

	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const la_lz4_seq *tab = table + table_off[bi];
	const uint8_t *s = src + b.src_off;
	const uint64_t s_room = src_bytes - b.src_off;	/* bytes of the image from s on */
	uint8_t *g_out = dst + doff;
	uint8_t *W = win + ((uintptr_t)g_out & 15);	/* W[i] <-> g_out[i], congruent mod 16 */
	volatile uint8_t *done_v = done;

	for (uint32_t k = tid; k < ns; k += FAST_THREADS) {
		dstpos[k] = tab[k].dst;
		done[k] = 0;
	}
	__syncthreads();

	const uint32_t nsteps = (ns + FAST_THREADS - 1) / FAST_THREADS;
	for (uint32_t r = 0; r < nsteps; r++) {
		const uint32_t k = (r * FAST_WAVES + wave) * 64 + lane;
		const bool active = k < ns;
		la_lz4_seq e = { 0, 0, 0, 0 };
		if (active)
			e = tab[k];
		const uint32_t d = e.dst, lit = e.lit_len, off = e.off;

		/* ---- literals: payload (global) -> window ---- */
		{
			const uint8_t *sp = s + e.lit_src;
			uint8_t *wp = W + d;
			uint32_t j = 0;
			while (j < lit) {
				uint32_t m = lit - j;
				if (m > 8) m = 8;
				uint64_t v;
				if ((uint64_t)e.lit_src + j + 8 <= s_room)
					v = ld_u64(sp + j);
				else {
					v = 0;
					for (uint32_t t = 0; t < m; t++)
						v |= (uint64_t)sp[j + t] << (8 * t);
				}
				for (uint32_t t = 0; t < m; t++)
					wp[j + t] = (uint8_t)(v >> (8 * t));
				j += m;
			}
		}

		const uint32_t mdst = d + lit;
		uint32_t next = olen;
		if (active && k + 1 < ns)
			next = dstpos[k + 1];
		const uint32_t mlen = active ? next - mdst : 0;
		bool fin = mlen == 0;
		if (active && fin) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			done_v[k] = 1;
		}

		/* ---- dependencies of the match: sequences [q, qend] ---- */
		uint32_t q = 1, qend = 0;
		const uint32_t s0 = mdst - off;
		if (!fin && s0 < d) {
			const uint32_t span = mlen < off ? mlen : off;
			uint32_t hi_byte = s0 + span - 1;
			if (hi_byte >= d)
				hi_byte = d - 1;
			/* largest index in [0, k-1] with dstpos[idx] <= x */
			uint32_t lo = 0, hi = k - 1;
			while (lo < hi) {
				uint32_t mid = (lo + hi + 1) >> 1;
				if (dstpos[mid] <= s0) lo = mid; else hi = mid - 1;
			}
			q = lo;
			qend = lo;
			while (qend + 1 < k && dstpos[qend + 1] <= hi_byte)
				qend++;
		}

		/* ---- match: wait for the overlapping earlier sequences, then copy ---- */
		for (;;) {
			if (!fin) {
				while (q <= qend && done_v[q])
					q++;
				if (q > qend) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					uint8_t *mp = W + mdst;
					const uint8_t *fp = W + s0;
					if (off >= mlen) {
						uint32_t i = 0;
						for (; i + 4 <= mlen; i += 4) {
							uint8_t a0 = fp[i], a1 = fp[i + 1], a2 = fp[i + 2], a3 = fp[i + 3];
							mp[i] = a0; mp[i + 1] = a1; mp[i + 2] = a2; mp[i + 3] = a3;
						}
						for (; i < mlen; i++)
							mp[i] = fp[i];
					} else {
						/* overlapping match: replicate with period `off`; every byte
						 * read is one this thread (or an earlier sequence) already wrote */
						for (uint32_t i = 0; i < mlen; i++)
							((volatile uint8_t *)mp)[i] = ((volatile uint8_t *)mp)[(int)i - (int)off];
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					done_v[k] = 1;
					fin = true;
				}
			}
			if (__ballot(!fin) == 0)
				break;
			__builtin_amdgcn_s_sleep(1);
		}
	}
	__syncthreads();

	/* ---- window -> decoded slab, 16 bytes per lane per step ---- */
	uint32_t head = (16u - (uint32_t)((uintptr_t)g_out & 15)) & 15u;
	if (head > olen) head = olen;
	if (tid < head)
		g_out[tid] = W[tid];
	const uint32_t nchunks = (olen - head) >> 4;
	const uint4 *wsrc = (const uint4 *)(W + head);
	uint4 *gdst = (uint4 *)(g_out + head);
	for (uint32_t c = tid; c < nchunks; c += FAST_THREADS)
		gdst[c] = wsrc[c];
	const uint32_t tail0 = head + (nchunks << 4);
	if (tail0 + tid < olen)
		g_out[tail0 + tid] = W[tail0 + tid];
}

void la_launch_lz4_expand_fast(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, const uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_expand_fast_kernel<LA_LZ4_FAST_MAXSEQ>, dim3(n), dim3(FAST_THREADS), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq,
	    d_table, d_table_off);
}
