/*
 * la_lz4_fast.hip -- LDS-window LZ4 expand kernel (gfx950): the hot kernel of
 * the lz4 filter path for independent blocks of at most 64 KiB
 * (libarchive/archive_read_support_filter_lz4.c:557-561, the
 * LZ4_decompress_safe call of BD=4 frames).
 *
 * One 512-thread workgroup owns one block; its whole output lives in a 64 KiB
 * LDS window (two workgroups per CU: 2 x 77 KiB of the 160 KiB LDS).  The
 * parse kernel has already reduced the token chain to
 *   - a table of sequences {literal source, literal length, output position,
 *     match offset}, and
 *   - a literal index: for every 16-byte chunk of the payload, the first
 *     sequence that still has literals in or after it,
 * so inside the block every copy is known up front:
 *
 *   phase L (literals)  one thread per 16-byte payload chunk: ONE coalesced
 *       16-byte load of the compressed stream, the chunk's sequences from the
 *       literal index, literal bytes scattered into the window.  Work is
 *       balanced by bytes, not by sequences; every payload byte is read once.
 *   phase M (matches)   one thread per sequence, sequences taken in increasing
 *       order.  A match may read bytes an earlier match produces, so every
 *       sequence publishes a "done" bit in LDS and a match waits only for the
 *       (typically one to three) earlier sequences that overlap its source
 *       range, found by a binary search over the output positions.
 *       Dependencies always point to lower sequence numbers and waves take
 *       sequences in increasing order, so the lowest unfinished sequence can
 *       always run: no deadlock, no barrier inside the phase.
 *   phase F (flush)     the window goes to the decoded slab with 16-byte
 *       coalesced stores (the window is placed so that LDS and HBM addresses
 *       are congruent modulo 16).
 *
 * HBM traffic per block: payload once, sequence table + literal index once in,
 * decoded bytes once out.
 */
#include "la_dev.h"

#define FAST_THREADS 512
#define FAST_WAVES   (FAST_THREADS / 64)

template <uint32_t MAXSEQ>
__global__ __launch_bounds__(FAST_THREADS) void lz4_expand_fast_kernel(
    const uint8_t *__restrict__ src, uint64_t src_bytes, const la_lz4_block *__restrict__ blocks,
    uint32_t n, uint8_t *__restrict__ dst, uint64_t dst_cap, const uint64_t *__restrict__ dst_off,
    const uint32_t *__restrict__ out_len, const uint32_t *__restrict__ status,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table,
    const uint64_t *__restrict__ table_off, const uint16_t *__restrict__ lit_index,
    const uint64_t *__restrict__ lidx_off)
{
	__shared__ __attribute__((aligned(16))) uint8_t win[65536 + 16];
	__shared__ uint16_t dstpos[MAXSEQ + 4];
	__shared__ uint32_t donebits[MAXSEQ / 32];

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
	const uint16_t *lidx = lit_index + lidx_off[bi];
	const uint8_t *s = src + b.src_off;
	const uint64_t s_room = src_bytes - b.src_off;	/* bytes of the image from s on */
	uint8_t *g_out = dst + doff;
	uint8_t *W = win + ((uintptr_t)g_out & 15);	/* W[i] <-> g_out[i], congruent mod 16 */
	volatile uint32_t *done_v = donebits;

	for (uint32_t k = tid; k < ns; k += FAST_THREADS)
		dstpos[k] = tab[k].dst;
	if (tid < MAXSEQ / 32)
		donebits[tid] = 0;

	/* ---- phase L: literals, one thread per 16-byte payload chunk ---- */
	const uint32_t nchunks = (b.src_len + 15) >> 4;
	for (uint32_t c = tid; c < nchunks; c += FAST_THREADS) {
		const uint32_t c0 = c << 4, c1 = c0 + 16;
		uint32_t k = lidx[c];
		uint4 v;
		if ((uint64_t)c1 <= s_room)
			v = ld_u128(s + c0);
		else {
			uint32_t t[4] = { 0, 0, 0, 0 };
			for (uint32_t i = 0; c0 + i < s_room && i < 16; i++)
				t[i >> 2] |= (uint32_t)s[c0 + i] << (8 * (i & 3));
			v = make_uint4(t[0], t[1], t[2], t[3]);
		}
		/* the first entries are fetched together: most chunks touch <= 3 sequences */
		la_lz4_seq e0 = { 0, 0, 0, 0 }, e1 = e0, e2 = e0;
		if (k < ns) e0 = tab[k];
		if (k + 1 < ns) e1 = tab[k + 1];
		if (k + 2 < ns) e2 = tab[k + 2];
		for (uint32_t it = 0; k < ns; it++, k++) {
			la_lz4_seq e = it == 0 ? e0 : it == 1 ? e1 : it == 2 ? e2 : tab[k];
			const uint32_t ls = e.lit_src, le = ls + e.lit_len;
			if (ls >= c1)
				break;
			const uint32_t lo = ls > c0 ? ls : c0, hi = le < c1 ? le : c1;
			for (uint32_t p = lo; p < hi; p++) {
				const uint32_t i = p - c0;
				const uint32_t dw = (i >> 2) == 0 ? v.x : (i >> 2) == 1 ? v.y : (i >> 2) == 2 ? v.z : v.w;
				W[e.dst + (p - ls)] = (uint8_t)(dw >> (8 * (i & 3)));
			}
			if (le >= c1)
				break;
		}
	}
	__syncthreads();

	/* ---- phase M: matches, one thread per sequence ---- */
	const uint32_t nsteps = (ns + FAST_THREADS - 1) / FAST_THREADS;
	for (uint32_t r = 0; r < nsteps; r++) {
		const uint32_t k = (r * FAST_WAVES + wave) * 64 + lane;
		const bool active = k < ns;
		la_lz4_seq e = { 0, 0, 0, 0 };
		if (active)
			e = tab[k];
		const uint32_t d = e.dst, off = e.off;
		const uint32_t mdst = d + e.lit_len;
		uint32_t next = olen;
		if (active && k + 1 < ns)
			next = dstpos[k + 1];
		const uint32_t mlen = active ? next - mdst : 0;
		bool fin = mlen == 0;
		if (active && fin)
			atomicOr(&donebits[k >> 5], 1u << (k & 31));

		/* sequences [q, qend] produce the bytes this match reads (those before its own sequence) */
		uint32_t q = 1, qend = 0;
		const uint32_t s0 = mdst - off;
		if (!fin && s0 < d) {
			const uint32_t span = mlen < off ? mlen : off;
			uint32_t hi_byte = s0 + span - 1;
			if (hi_byte >= d)
				hi_byte = d - 1;
			uint32_t lo = 0, hi = k - 1;	/* largest index with dstpos[idx] <= s0 */
			while (lo < hi) {
				uint32_t mid = (lo + hi + 1) >> 1;
				if (dstpos[mid] <= s0) lo = mid; else hi = mid - 1;
			}
			q = lo;
			qend = lo;
			while (qend + 1 < k && dstpos[qend + 1] <= hi_byte)
				qend++;
		}

		for (;;) {
			if (!fin) {
				/* advance q over finished sequences, a 32-bit word of flags at a time */
				while (q <= qend) {
					/* bit 0 of `word` is the flag of q; the zeros shifted in from the
					 * top end the run at the word boundary */
					uint32_t word = done_v[q >> 5] >> (q & 31);
					uint32_t inv = ~word;
					uint32_t run = inv ? (uint32_t)__builtin_ctz(inv) : 32u;
					if (run == 0)
						break;
					q += run;
				}
				if (q > qend) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					uint8_t *mp = W + mdst;
					const uint8_t *fp = W + s0;
					if (off >= mlen) {
						uint32_t i = 0;
						for (; i + 8 <= mlen; i += 8) {
							uint8_t a0 = fp[i], a1 = fp[i + 1], a2 = fp[i + 2], a3 = fp[i + 3];
							uint8_t a4 = fp[i + 4], a5 = fp[i + 5], a6 = fp[i + 6], a7 = fp[i + 7];
							mp[i] = a0; mp[i + 1] = a1; mp[i + 2] = a2; mp[i + 3] = a3;
							mp[i + 4] = a4; mp[i + 5] = a5; mp[i + 6] = a6; mp[i + 7] = a7;
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
					atomicOr(&donebits[k >> 5], 1u << (k & 31));
					fin = true;
				}
			}
			if (__ballot(!fin) == 0)
				break;
			__builtin_amdgcn_s_sleep(1);
		}
	}
	__syncthreads();

	/* ---- phase F: window -> decoded slab, 16 bytes per lane per step ---- */
	uint32_t head = (16u - (uint32_t)((uintptr_t)g_out & 15)) & 15u;
	if (head > olen) head = olen;
	if (tid < head)
		g_out[tid] = W[tid];
	const uint32_t nflush = (olen - head) >> 4;
	const uint4 *wsrc = (const uint4 *)(W + head);
	uint4 *gdst = (uint4 *)(g_out + head);
	for (uint32_t c = tid; c < nflush; c += FAST_THREADS)
		gdst[c] = wsrc[c];
	const uint32_t tail0 = head + (nflush << 4);
	if (tail0 + tid < olen)
		g_out[tail0 + tid] = W[tail0 + tid];
}

void la_launch_lz4_expand_fast(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, const uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off,
    const uint16_t *d_lidx, const uint64_t *d_lidx_off)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_expand_fast_kernel<LA_LZ4_FAST_MAXSEQ>, dim3(n), dim3(FAST_THREADS), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq,
	    d_table, d_table_off, d_lidx, d_lidx_off);
}
