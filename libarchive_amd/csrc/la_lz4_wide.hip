/*
 * la_lz4_wide.hip -- LZ4 expand kernel with the output window in GLOBAL memory (gfx950): third
 * implementation of the expand step of independent blocks with a sequence table
 * (libarchive/archive_read_support_filter_lz4.c:557-561, the LZ4_decompress_safe call), selected
 * with LA_LZ4_OPT_EXPAND_WIDE.  Same results as the LDS-window kernels; NOT the default (measured
 * slower on C2: DESIGN.md section 5b), kept as a cross-check and as the record of the experiment.
 *
 * The LDS-window kernels keep a block's 64 KiB of output in LDS, so a CU holds TWO blocks, and a block
 * advances one level of its match dependency DAG per poll iteration with nothing to hide the latency
 * behind.  This kernel gives the window up: ONE WAVE owns a block, writes its output straight into
 * the decoded slab and reads match sources back from there (L1 / L2 / Infinity Cache), so a CU holds
 * as many blocks as its LDS bookkeeping allows (10) and one block's round trips hide behind the others'.
 *
 *   pass 1   64 sequences at a time in stream order, one per lane: the literal run payload -> slab
 *            (16-byte pieces, whole dwords), and the LEVEL of every match = 1 + the highest
 *            completion level among the sequences its source bytes lie in (two lower bounds over the
 *            sequence ends in LDS; dependencies inside the group resolve in ballot rounds).  An
 *            overlapping match (offset < length) is copied in pieces of a doubling period and
 *            completes that many levels later.
 *   sort     counting sort of the matches by level (LDS atomics, one wave scan).
 *   pass 2   the levels in order: free lanes take the level's matches, copy slab -> slab with 16-byte
 *            pieces (the last one end-aligned), then ONE fence (s_waitcnt vmcnt(0): the wave's stores
 *            are in L2 / its own L1 before its next loads).  A lane keeps an overlapping match until its
 *            last piece.  C2 blocks: about 2030 sequences in 23 to 34 levels.
 *   in-order blocks with more sequences than the LDS arrays hold, or more levels than the counters:
 *            64 sequences at a time, the group's matches in ballot rounds with a fence each.
 * No barrier between waves (one wave per workgroup); every loop is bounded by the sequence count.
 */
#include "la_dev.h"

typedef uint64_t seq_t;
#define SEQ_LIT_SRC(e) ((uint32_t)((e) & 0xFFFFu))
#define SEQ_LIT_LEN(e) ((uint32_t)(((e) >> 16) & 0xFFFFu))
#define SEQ_DST(e)     ((uint32_t)(((e) >> 32) & 0xFFFFu))
#define SEQ_OFF(e)     ((uint32_t)((e) >> 48))

__device__ __forceinline__ uint4 g_ld16(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void g_st16(uint8_t *p, uint4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint64_t g_ld8(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void g_st8(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void g_st4(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

__device__ __forceinline__ uint32_t g_ld4(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

/* n bytes d := s, ranges disjoint: nothing outside [d, d + n) written, nothing outside [s, s + n) read */
__device__ __forceinline__ void copy_exact(uint8_t *d, const uint8_t *s, uint32_t n)
{
	uint32_t i = 0;
	for (; i + 32u <= n; i += 32) {
		const uint4 a = g_ld16(s + i), b = g_ld16(s + i + 16);
		g_st16(d + i, a);
		g_st16(d + i + 16, b);
	}
	const uint32_t r = n - i;	/* 0..31 */
	if (r >= 16u) {
		const uint4 a = g_ld16(s + i), t = g_ld16(s + n - 16);
		g_st16(d + i, a);
		if (r != 16u)
			g_st16(d + n - 16, t);
	} else if (r >= 8u) {
		const uint64_t a = g_ld8(s + i), t = g_ld8(s + n - 8);
		g_st8(d + i, a);
		if (r != 8u)
			g_st8(d + n - 8, t);
	} else if (r >= 4u) {
		const uint32_t a = g_ld4(s + i), t = g_ld4(s + n - 4);
		g_st4(d + i, a);
		if (r != 4u)
			g_st4(d + n - 4, t);
	} else if (r) {
		const uint8_t a = s[i], b = r > 1u ? s[i + 1] : (uint8_t)0, c = r > 2u ? s[i + 2] : (uint8_t)0;
		d[i] = a;
		if (r > 1u) d[i + 1] = b;
		if (r > 2u) d[i + 2] = c;
	}
}

__device__ __forceinline__ void wave_fence()
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/* One sequence's literal run payload -> slab, whole dwords (the spill of up to three bytes lands in the lane's own
 * match area; a sequence without a match of four bytes behind it is copied exactly). */
__device__ __forceinline__ void copy_literals(uint8_t *op, const uint8_t *lp, uint32_t ls, uint32_t ll, uint32_t mlen,
    uint64_t s_room, bool have)
{
	const bool safe = (uint64_t)ls + ll + 16u <= s_room;	/* 16-byte loads may run past the run, never past the image */
	uint32_t nb = have ? ll : 0u;
	if (nb && mlen >= 4u && safe)
		nb = (nb + 3u) & ~3u;
	if (have && ll && !safe) {	/* the image's last bytes: one by one */
		for (uint32_t i = 0; i < ll; i++)
			op[i] = (uint64_t)ls + i < s_room ? lp[i] : (uint8_t)0;
		nb = 0;
	}
	uint32_t i = 0;
	for (; i + 16u <= nb; i += 16)
		g_st16(op + i, g_ld16(lp + i));
	const uint32_t r = nb - i;	/* 0..15 */
	if (r) {
		const uint4 v = g_ld16(lp + i);	/* (safe: checked above) */
		if (r >= 8u) {
			g_st8(op + i, (uint64_t)v.x | ((uint64_t)v.y << 32));
			if (r >= 12u) {
				g_st4(op + i + 8, v.z);
				if (r > 12u) {	/* 13..15: exact run */
					const uint32_t w = v.w;
					op[i + 12] = (uint8_t)w;
					if (r > 13u) op[i + 13] = (uint8_t)(w >> 8);
					if (r > 14u) op[i + 14] = (uint8_t)(w >> 16);
				}
			} else if (r > 8u) {	/* 9..11 */
				const uint32_t w = v.z;
				op[i + 8] = (uint8_t)w;
				if (r > 9u) op[i + 9] = (uint8_t)(w >> 8);
				if (r > 10u) op[i + 10] = (uint8_t)(w >> 16);
			}
		} else if (r >= 4u) {
			g_st4(op + i, v.x);
			if (r > 4u) {	/* 5..7 */
				const uint32_t w = v.y;
				op[i + 4] = (uint8_t)w;
				if (r > 5u) op[i + 5] = (uint8_t)(w >> 8);
				if (r > 6u) op[i + 6] = (uint8_t)(w >> 16);
			}
		} else {
			const uint32_t w = v.x;
			op[i] = (uint8_t)w;
			if (r > 1u) op[i + 1] = (uint8_t)(w >> 8);
			if (r > 2u) op[i + 2] = (uint8_t)(w >> 16);
		}
	}
}

/* next piece of a match: a whole match when offset >= length, otherwise pieces of a doubling multiple of the period */
__device__ __forceinline__ void match_piece(uint8_t *out, uint32_t mdst, uint32_t off, uint32_t mlen, uint32_t &done, uint32_t &eff)
{
	if (2u * eff <= off + done)
		eff *= 2u;	/* a multiple of the period that is already in place */
	uint32_t nb = mlen - done;
	if (nb > eff) nb = eff;
	copy_exact(out + mdst + done, out + mdst + done - eff, nb);
	done += nb;
}

/* Blocks the level-sorted path does not take (more sequences than its LDS arrays hold, or a dependency chain
 * longer than its level table): 64 sequences at a time in stream order, the group's matches in rounds. */
__device__ void wide_in_order(const uint64_t *tab, uint32_t ns, uint32_t olen, const uint8_t *s, uint64_t s_room, uint8_t *out, uint32_t lane)
{
	for (uint32_t kb = 0; kb < ns; kb += 64) {
		const uint32_t k = kb + lane;
		const bool have = k < ns;
		const seq_t e = have ? tab[k] : 0;
		const uint32_t next_dst = (have && k + 1 < ns) ? SEQ_DST((seq_t)tab[k + 1]) : olen;
		const uint32_t d = SEQ_DST(e), ll = SEQ_LIT_LEN(e), off = SEQ_OFF(e), ls = SEQ_LIT_SRC(e);
		const uint32_t mdst = d + ll;
		const uint32_t mlen = have ? next_dst - mdst : 0u;
		const uint32_t end = have ? next_dst : 0xFFFFFFFFu;	/* first byte behind this sequence (increasing over the lanes) */

		copy_literals(out + d, s + ls, ls, ll, mlen, s_room, have);
		wave_fence();

		bool pendm = have && mlen != 0u && off != 0u && off <= mdst;
		const uint32_t s0 = mdst - off;
		const uint32_t span = mlen < off ? mlen : off;
		/* lanes of this group whose sequences the source touches: [jlo, jhi] by binary search over `end` */
		uint64_t depmask = 0;
		{
			const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);	/* first output byte of the group */
			const uint32_t shi = s0 + span - 1u;
			uint32_t jlo = 0, jhi = 0;
#pragma unroll
			for (uint32_t bit = 32; bit; bit >>= 1) {
				/* number of lanes whose sequence ends at or before s0 / shi */
				const uint32_t e_lo = (uint32_t)__shfl((int)end, (int)(jlo + bit - 1u), 64);
				if (e_lo <= s0) jlo += bit;
				const uint32_t e_hi = (uint32_t)__shfl((int)end, (int)(jhi + bit - 1u), 64);
				if (e_hi <= shi) jhi += bit;
			}
			if (pendm && shi >= g0 && lane != 0) {
				const uint32_t hi = jhi < lane ? jhi : lane - 1u;	/* own literals are in place */
				if (jlo <= hi) {
					const uint64_t upto = hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull);
					depmask = upto & ~((1ull << jlo) - 1ull);
				}
			}
		}
		uint32_t done = 0, eff = off;
		for (;;) {
			const uint64_t pending = __ballot(pendm);
			if (pending == 0)
				break;
			if (pendm && (depmask & pending) == 0) {
				match_piece(out, mdst, off, mlen, done, eff);
				if (done == mlen)
					pendm = false;
				depmask = 0;	/* (the next piece only waits for this one, which the fence covers) */
			}
			wave_fence();
		}
	}
}

#define WCAP  2560u	/* sequences per block the level-sorted path holds in LDS */
#define WLMAX 1023u	/* dependency levels it sorts by */

__global__ __launch_bounds__(64) void lz4_expand_wide_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    const la_lz4_block *__restrict__ blocks, uint32_t n, uint8_t *dst, uint64_t dst_cap,
    const uint64_t *__restrict__ dst_off, const uint32_t *__restrict__ out_len, uint32_t *status_out,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table, const uint64_t *__restrict__ table_off,
    uint32_t long_thr)
{
	__shared__ uint16_t s_end[WCAP];	/* level pass: last byte of every sequence; afterwards: the sequences sorted by level */
	__shared__ uint16_t s_lv[WCAP];		/* level a sequence's match starts in (0: no match) */
	__shared__ uint32_t s_dl32[WCAP / 2u];	/* level pass: u16 level a sequence's match is COMPLETE in; afterwards: u32 per-level counters */
	uint16_t *s_dl = (uint16_t *)s_dl32;

	const uint32_t lane = threadIdx.x;
	const uint32_t bi = blockIdx.x;
	if (bi >= n)
		return;
	const la_lz4_block b = blocks[bi];
	const uint32_t olen = out_len[bi];
	const uint32_t ns = nseq[bi];
	const uint64_t doff = dst_off[bi];
	/* same predicate as the LDS-window kernels (0xFFFFFFFF: the block has no table) */
	if (status_out[bi] != LA_ST_OK || olen == 0 || !la_lz4_fast_eligible(b) || ns == 0xFFFFFFFFu ||
	    doff + olen > dst_cap || la_lz4_long_sequences(ns, olen, long_thr))
		return;
	const uint64_t *tab = (const uint64_t *)(const void *)(table + table_off[bi]);
	const uint8_t *s = src + b.src_off;
	const uint64_t s_room = src_bytes - b.src_off;
	uint8_t *out = dst + doff;
	if (ns > WCAP) {
		wide_in_order(tab, ns, olen, s, s_room, out, lane);
		return;
	}

	/* ---- pass 1, 64 sequences at a time in stream order: literals out, and the level of every match =
	 * 1 + the highest completion level among the sequences its source reaches into ---- */
	uint32_t my_max = 0;
	for (uint32_t kb = 0; kb < ns; kb += 64) {
		const uint32_t k = kb + lane;
		const bool have = k < ns;
		const seq_t e = have ? tab[k] : 0;
		const uint32_t next_dst = (have && k + 1 < ns) ? SEQ_DST((seq_t)tab[k + 1]) : olen;
		const uint32_t d = SEQ_DST(e), ll = SEQ_LIT_LEN(e), off = SEQ_OFF(e), ls = SEQ_LIT_SRC(e);
		const uint32_t mdst = d + ll;
		const uint32_t mlen = have ? next_dst - mdst : 0u;
		if (have)
			s_end[k] = (uint16_t)(next_dst - 1u);
		copy_literals(out + d, s + ls, ls, ll, mlen, s_room, have);
		__syncthreads();

		const bool hasm = have && mlen != 0u && off != 0u && off <= mdst;
		const uint32_t s0 = mdst - off;
		const uint32_t shi = s0 + (mlen < off ? mlen : off) - 1u;
		/* sequences [jlo, jhi] hold the source bytes: lower bounds over the ends of the sequences before this one */
		uint32_t jlo = 0, jhi = 0;
		if (hasm) {
#pragma unroll
			for (uint32_t step = 2048; step; step >>= 1) {
				const uint32_t pl = jlo + step, ph = jhi + step;
				if (pl <= k && (uint32_t)s_end[pl - 1u] < s0) jlo = pl;
				if (ph <= k && (uint32_t)s_end[ph - 1u] < shi) jhi = ph;
			}
		}
		const bool nodeps = !hasm || jlo >= k;	/* (the source lies in the sequence's own literals) */
		if (jhi >= k) jhi = k - 1u;		/* (only read when !nodeps, and then k >= 1) */
		/* pieces of an overlapping match (the same recurrence as match_piece) */
		uint32_t pieces = 1;
		if (hasm && off < mlen) {
			uint32_t dn = 0, ef = off;
			pieces = 0;
			while (dn < mlen) {
				if (2u * ef <= off + dn) ef *= 2u;
				dn += (mlen - dn < ef) ? mlen - dn : ef;
				pieces++;
			}
		}
		bool known = !hasm;
		if (have && !hasm) {
			s_lv[k] = 0;
			s_dl[k] = 0;
		}
		/* dependencies inside the group: wait for those lanes' levels */
		uint64_t inmask = 0;
		if (!nodeps && jhi >= kb) {
			const uint32_t a = jlo > kb ? jlo - kb : 0u, z = jhi - kb;	/* z < lane */
			inmask = ((z >= 63u) ? ~0ull : ((1ull << (z + 1u)) - 1ull)) & ~((1ull << a) - 1ull);
		}
		for (;;) {
			const uint64_t pend = __ballot(!known);
			if (pend == 0)
				break;
			if (!known && (inmask & pend) == 0) {
				uint32_t m = 0;
				if (!nodeps)
					for (uint32_t j = jlo; j <= jhi; j++) {
						const uint32_t v = s_dl[j];
						m = v > m ? v : m;
					}
				const uint32_t lv = m + 1u, dl = m + pieces;
				s_lv[k] = (uint16_t)(lv > 0xFFFFu ? 0xFFFFu : lv);
				s_dl[k] = (uint16_t)(dl > 0xFFFFu ? 0xFFFFu : dl);
				my_max = dl > my_max ? dl : my_max;
				known = true;
			}
			__syncthreads();
		}
	}
	wave_fence();	/* the literals are in place */
	uint32_t maxl = my_max;
#pragma unroll
	for (int sh = 32; sh; sh >>= 1) {
		const uint32_t o = (uint32_t)__shfl_xor((int)maxl, sh, 64);
		maxl = o > maxl ? o : maxl;
	}
	maxl = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxl);
	if (maxl > WLMAX) {
		__syncthreads();
		wide_in_order(tab, ns, olen, s, s_room, out, lane);
		return;
	}

	/* ---- counting sort of the matches by level (s_dl32 becomes the counters, s_end the sorted list) ---- */
	__syncthreads();
	for (uint32_t i = lane; i <= maxl + 1u; i += 64)
		s_dl32[i] = 0;
	__syncthreads();
	for (uint32_t k = lane; k < ns; k += 64) {
		const uint32_t lv = s_lv[k];
		if (lv)
			atomicAdd(&s_dl32[lv], 1u);
	}
	__syncthreads();
	{
		uint32_t base = 0;	/* uniform */
		for (uint32_t c = 0; c <= maxl; c += 64) {
			const uint32_t v = (c + lane <= maxl) ? s_dl32[c + lane] : 0u;
			uint32_t incl = v;
#pragma unroll
			for (int sh = 1; sh < 64; sh <<= 1) {
				const uint32_t o = (uint32_t)__shfl_up((int)incl, sh, 64);
				if ((int)lane >= sh) incl += o;
			}
			if (c + lane <= maxl)
				s_dl32[c + lane] = base + incl - v;	/* first slot of the level */
			base += (uint32_t)__shfl((int)incl, 63, 64);
		}
	}
	__syncthreads();
	for (uint32_t k = lane; k < ns; k += 64) {
		const uint32_t lv = s_lv[k];
		if (lv) {
			const uint32_t pos = atomicAdd(&s_dl32[lv], 1u);
			s_end[pos] = (uint16_t)k;
		}
	}
	__syncthreads();
	/* now s_dl32[l] = one behind the last slot of level l (s_dl32[0] = 0) */

	/* ---- pass 2: the levels in order, one fence per level; a lane keeps an overlapping match until its last piece ---- */
	bool busy = false;
	uint32_t mdst = 0, off = 1, mlen = 0, done = 0, eff = 1;
	uint32_t pos = 0, r = 1;	/* uniform */
	for (;;) {
		const uint32_t lvl_end = r <= maxl ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s_dl32[r]) : pos;
		if (busy) {
			match_piece(out, mdst, off, mlen, done, eff);
			if (done == mlen)
				busy = false;
		}
		while (pos < lvl_end) {
			const uint64_t fr = __ballot(!busy);
			const uint32_t nf = (uint32_t)__popcll(fr);
			if (nf == 0)
				break;
			const uint32_t avail = lvl_end - pos;
			const uint32_t take = nf < avail ? nf : avail;
			const uint32_t rank = (uint32_t)__popcll(fr & ((1ull << lane) - 1ull));
			if (!busy && rank < take) {
				const uint32_t k = s_end[pos + rank];
				const seq_t e = tab[k];
				const uint32_t next_dst = (k + 1 < ns) ? SEQ_DST((seq_t)tab[k + 1]) : olen;
				mdst = SEQ_DST(e) + SEQ_LIT_LEN(e);
				off = SEQ_OFF(e);
				mlen = next_dst - mdst;
				done = 0;
				eff = off;
				match_piece(out, mdst, off, mlen, done, eff);
				busy = done != mlen;
			}
			pos += take;
		}
		wave_fence();
		if (pos == lvl_end) {
			if (r > maxl) {
				if (__ballot(busy) == 0)
					break;
			} else {
				r++;
			}
		}
	}
}

void la_launch_lz4_expand_wide(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t long_thr)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_expand_wide_kernel, dim3(n), dim3(64), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq, d_table, d_table_off, long_thr);
}

/* =====================================================================================================================
 * lz4_expand_ring_kernel (LA_LZ4_OPT_EXPAND_RING): one wave per block, 64 sequences per group in stream order like the
 * in-order path above, but the group's own dependencies are resolved in LDS: the wave keeps the last 8 KiB of the
 * block's output in a ring.  A match whose source lies in the ring reads it there (its in-group producers are waited
 * for in ballot rounds that cost LDS latency, not a fence: LDS operations of a wave run in order); a match whose source
 * lies wholly in front of the group reads the slab (those stores were fenced at the start of the group, and the loads
 * are issued before the rounds so that they overlap them).  Every byte goes to the slab and to the ring.  ONE
 * s_waitcnt vmcnt(0) per group.  Groups that do not fit the scheme (more than 4 KiB of output, or a source that
 * straddles the ring's lower edge and the group) take the fenced rounds of the in-order path.
 * Measured: 9.65 ms per 4 GiB of C2 against 5.4 for the LDS-window kernel (profiles/r02_wide_kernel.txt); a cross-check.
 * ===================================================================================================================== */
#define RING_BYTES 8192u
#define RING_MASK  (RING_BYTES - 1u)

/* ring accesses: position p of the block's output lives at ring[p & RING_MASK]; an access that would run over the
 * ring's end goes byte by byte */
__device__ __forceinline__ uint4 ring_ld16(const uint8_t *ring, uint32_t p)
{
	const uint32_t i = p & RING_MASK;
	uint4 v;
	if (i + 16u <= RING_BYTES) { __builtin_memcpy(&v, ring + i, 16); return v; }
	uint64_t lo = 0, hi = 0;	/* (no byte arrays: they would live in scratch memory) */
	for (uint32_t k = 0; k < 8; k++) {
		lo |= (uint64_t)ring[(p + k) & RING_MASK] << (8 * k);
		hi |= (uint64_t)ring[(p + 8 + k) & RING_MASK] << (8 * k);
	}
	v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32); v.z = (uint32_t)hi; v.w = (uint32_t)(hi >> 32);
	return v;
}
/* the first n (1..16) bytes of v to q[0..n): 8 / 4 / 2 / 1 pieces out of registers */
__device__ __forceinline__ void put_n(uint8_t *q, uint4 v, uint32_t n)
{
	uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
	const uint64_t hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
	if (n & 16u) { __builtin_memcpy(q, &v, 16); return; }
	if (n & 8u) { __builtin_memcpy(q, &lo, 8); q += 8; lo = hi; }
	if (n & 4u) { const uint32_t w = (uint32_t)lo; __builtin_memcpy(q, &w, 4); q += 4; lo >>= 32; }
	if (n & 2u) { const uint16_t w = (uint16_t)lo; __builtin_memcpy(q, &w, 2); q += 2; lo >>= 16; }
	if (n & 1u) *q = (uint8_t)lo;
}
__device__ __forceinline__ void ring_st(uint8_t *ring, uint32_t p, uint4 v, uint32_t n /* 1..16 */)
{
	const uint32_t i = p & RING_MASK;
	if (i + n <= RING_BYTES) { put_n(ring + i, v, n); return; }
	uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
	for (uint32_t k = 0; k < n; k++) {
		ring[(p + k) & RING_MASK] = (uint8_t)lo;
		lo = (lo >> 8) | (hi << 56); hi >>= 8;
	}
}
__device__ __forceinline__ void slab_st(uint8_t *q, uint4 v, uint32_t n /* 1..16 */) { put_n(q, v, n); }

/* one group with fenced rounds through the slab (the in-order path's body for 64 sequences); nothing goes to the ring */
__device__ void ring_group_fenced(uint8_t *out, uint32_t lane, bool have, uint32_t d, uint32_t ll, uint32_t off, uint32_t mdst,
    uint32_t mlen, uint32_t end)
{
	(void)ll;
	bool pendm = have && mlen != 0u && off != 0u && off <= mdst;
	const uint32_t s0 = mdst - off;
	const uint32_t span = mlen < off ? mlen : off;
	uint64_t depmask = 0;
	{
		const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
		const uint32_t shi = s0 + span - 1u;
		uint32_t jlo = 0, jhi = 0;
#pragma unroll
		for (uint32_t bit = 32; bit; bit >>= 1) {
			const uint32_t e_lo = (uint32_t)__shfl((int)end, (int)(jlo + bit - 1u), 64);
			if (e_lo <= s0) jlo += bit;
			const uint32_t e_hi = (uint32_t)__shfl((int)end, (int)(jhi + bit - 1u), 64);
			if (e_hi <= shi) jhi += bit;
		}
		if (pendm && shi >= g0 && lane != 0) {
			const uint32_t hi = jhi < lane ? jhi : lane - 1u;
			if (jlo <= hi) {
				const uint64_t upto = hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull);
				depmask = upto & ~((1ull << jlo) - 1ull);
			}
		}
	}
	uint32_t done = 0, eff = off;
	for (;;) {
		const uint64_t pending = __ballot(pendm);
		if (pending == 0)
			break;
		if (pendm && (depmask & pending) == 0) {
			match_piece(out, mdst, off, mlen, done, eff);
			if (done == mlen)
				pendm = false;
			depmask = 0;
		}
		wave_fence();
	}
}

__global__ __launch_bounds__(64) void lz4_expand_ring_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    const la_lz4_block *__restrict__ blocks, uint32_t n, uint8_t *dst, uint64_t dst_cap,
    const uint64_t *__restrict__ dst_off, const uint32_t *__restrict__ out_len, uint32_t *status_out,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table, const uint64_t *__restrict__ table_off,
    uint32_t long_thr)
{
	__shared__ uint8_t ring[RING_BYTES];
	const uint32_t lane = threadIdx.x;
	const uint32_t bi = blockIdx.x;
	if (bi >= n)
		return;
	const la_lz4_block b = blocks[bi];
	const uint32_t olen = out_len[bi];
	const uint32_t ns = nseq[bi];
	const uint64_t doff = dst_off[bi];
	if (status_out[bi] != LA_ST_OK || olen == 0 || !la_lz4_fast_eligible(b) || ns == 0xFFFFFFFFu ||
	    doff + olen > dst_cap || la_lz4_long_sequences(ns, olen, long_thr))
		return;
	const uint64_t *tab = (const uint64_t *)(const void *)(table + table_off[bi]);
	const uint8_t *s = src + b.src_off;
	const uint64_t s_room = src_bytes - b.src_off;
	uint8_t *out = dst + doff;
	uint32_t ring_from = 0;	/* output positions >= ring_from (and within RING_BYTES of the newest byte) are in the ring */

	for (uint32_t kb = 0; kb < ns; kb += 64) {
		const uint32_t k = kb + lane;
		const bool have = k < ns;
		const seq_t e = have ? tab[k] : 0;
		const uint32_t next_dst = (have && k + 1 < ns) ? SEQ_DST((seq_t)tab[k + 1]) : olen;
		const uint32_t d = SEQ_DST(e), ll = SEQ_LIT_LEN(e), off = SEQ_OFF(e), ls = SEQ_LIT_SRC(e);
		const uint32_t mdst = d + ll;
		const uint32_t mlen = have ? next_dst - mdst : 0u;
		const uint32_t end = have ? next_dst : 0xFFFFFFFFu;
		const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
		const uint32_t last = (ns - kb < 64u ? ns - kb : 64u) - 1u;
		const uint32_t g1 = (uint32_t)__builtin_amdgcn_readlane((int)next_dst, __builtin_amdgcn_readfirstlane((int)last));	/* first byte behind the group */
		const bool hasm = have && mlen != 0u && off != 0u && off <= mdst;
		const uint32_t s0 = mdst - off;
		const uint32_t span = mlen < off ? mlen : off;
		const uint32_t ring_lo = (g1 > RING_BYTES && g1 - RING_BYTES > ring_from) ? g1 - RING_BYTES : ring_from;
		const bool in_ring = hasm && s0 >= ring_lo;			/* the whole source is (or will be) in the ring */
		const bool in_slab = hasm && !in_ring && s0 + span <= g0;	/* the whole source lies in front of the group */
		const bool fits = (g1 - g0) <= RING_BYTES / 2u && __ballot(hasm && !in_ring && !in_slab) == 0;

		wave_fence();	/* everything the earlier groups stored is in the slab */
		if (!fits) {
			copy_literals(out + d, s + ls, ls, ll, mlen, s_room, have);
			wave_fence();
			ring_group_fenced(out, lane, have, d, ll, off, mdst, mlen, end);
			ring_from = g1;	/* (nothing of this group is in the ring) */
			continue;
		}

		/* literals: payload -> slab and ring (exact lengths in the ring: no spill) */
		{
			const uint8_t *lp = s + ls;
			const bool safe = (uint64_t)ls + ll + 16u <= s_room;
			const uint32_t nb = have ? ll : 0u;
			for (uint32_t i = 0; i < nb; i += 16) {
				const uint32_t m = nb - i < 16u ? nb - i : 16u;
				uint4 v;
				if (safe) v = g_ld16(lp + i);
				else {	/* the image's last bytes */
					uint64_t lo = 0, hi = 0;
					for (uint32_t q = 0; q < m; q++) {
						const uint64_t c = (uint64_t)ls + i + q < s_room ? lp[i + q] : (uint8_t)0;
						if (q < 8) lo |= c << (8 * q); else hi |= c << (8 * (q - 8));
					}
					v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32); v.z = (uint32_t)hi; v.w = (uint32_t)(hi >> 32);
				}
				slab_st(out + d + i, v, m);
				ring_st(ring, d + i, v, m);
			}
		}

		/* matches from the slab: no dependency inside the group (the stores they read were fenced above) */
		if (in_slab) {
			const uint8_t *sp = out + s0;
			if (off >= mlen && mlen >= 16u) {
				uint32_t i = 0;
				for (; i + 16u <= mlen; i += 16) {
					const uint4 v = g_ld16(sp + i);
					g_st16(out + mdst + i, v);
					ring_st(ring, mdst + i, v, 16u);
				}
				if (mlen & 15u) {	/* last piece end-aligned (rewrites some bytes with the same values) */
					const uint4 v = g_ld16(sp + mlen - 16u);
					g_st16(out + mdst + mlen - 16u, v);
					ring_st(ring, mdst + mlen - 16u, v, 16u);
				}
			} else {	/* short, or overlapping with its whole period in front of the group: byte k is source byte k mod off */
				uint32_t m = 0;
				for (uint32_t i = 0; i < mlen; i++) {
					const uint8_t c = sp[m];
					out[mdst + i] = c;
					ring[(mdst + i) & RING_MASK] = c;
					m = m + 1u == off ? 0u : m + 1u;
				}
			}
		}

		/* matches from the ring: wait for the lanes of this group whose sequences the source touches */
		bool pendm = in_ring;
		uint64_t depmask = 0;
		{
			const uint32_t shi = s0 + span - 1u;
			uint32_t jlo = 0, jhi = 0;
#pragma unroll
			for (uint32_t bit = 32; bit; bit >>= 1) {
				const uint32_t e_lo = (uint32_t)__shfl((int)end, (int)(jlo + bit - 1u), 64);
				if (e_lo <= s0) jlo += bit;
				const uint32_t e_hi = (uint32_t)__shfl((int)end, (int)(jhi + bit - 1u), 64);
				if (e_hi <= shi) jhi += bit;
			}
			if (pendm && shi >= g0 && lane != 0) {
				const uint32_t hi = jhi < lane ? jhi : lane - 1u;	/* own literals are in place */
				if (jlo <= hi) {
					const uint64_t upto = hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull);
					depmask = upto & ~((1ull << jlo) - 1ull);
				}
			}
		}
		/* (in_slab matches count as pending producers until here: they are done now) */
		for (;;) {
			const uint64_t pending = __ballot(pendm);
			if (pending == 0)
				break;
			if (pendm && (depmask & pending) == 0) {
				if (off >= 16u || off >= mlen) {
					/* pieces of 16 bytes in order: a piece never reads bytes a later piece writes, and LDS runs in order */
					for (uint32_t i = 0; i < mlen; i += 16) {
						const uint32_t m = mlen - i < 16u ? mlen - i : 16u;
						const uint4 v = ring_ld16(ring, s0 + i);
						ring_st(ring, mdst + i, v, m);
						slab_st(out + mdst + i, v, m);
					}
				} else {	/* short period: byte by byte in the ring, then the slab in pieces */
					for (uint32_t i = 0; i < mlen; i++)
						ring[(mdst + i) & RING_MASK] = ring[(s0 + i) & RING_MASK];
					for (uint32_t i = 0; i < mlen; i += 16) {
						const uint32_t m = mlen - i < 16u ? mlen - i : 16u;
						slab_st(out + mdst + i, ring_ld16(ring, mdst + i), m);
					}
				}
				pendm = false;
			}
		}
	}
}

void la_launch_lz4_expand_ring(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t long_thr)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_expand_ring_kernel, dim3(n), dim3(64), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq, d_table, d_table_off, long_thr);
}
