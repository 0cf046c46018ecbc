/*
 * la_lz4_fastq.hip -- LDS-window LZ4 expand kernel (gfx950), QUEUE generation: the second
 * implementation of the expand step for independent blocks of at most 64 KiB
 * (selected with LA_LZ4_OPT_EXPAND_QUEUE; the default is la_lz4_fast.hip)
 * (libarchive/archive_read_support_filter_lz4.c:557-561, the
 * LZ4_decompress_safe call of BD=4 frames).
 *
 * One 512-thread workgroup owns one block; its whole output lives in a 64 KiB
 * LDS window (two workgroups per CU).  The parse kernel has already reduced the
 * token chain to a table of sequences {literal source, literal length, output
 * position, match offset}, so inside the block every copy is known up front:
 *
 *   prepass             every thread loads its sequences' entries, marks their
 *       LITERAL bytes valid in a byte-granular bitmap of the window (one bit per
 *       output byte, 8 KiB) and builds the chunk index: for every 32-byte chunk
 *       of the payload, the first sequence that still has literals in or after it.
 *   phase L (literals)  one thread per 32-byte payload chunk: two coalesced
 *       16-byte loads of the compressed stream, the chunk's sequences from the
 *       chunk index, literal bytes into the window as whole dwords with static
 *       register indices.  Work is balanced by bytes, not by sequences; every
 *       payload byte is read once.
 *   phase M (matches)   rounds.  In a round every thread looks at the bitmap words
 *       under the SOURCE of each of its unfinished matches (one aligned 8-byte
 *       LDS read; how far the source is valid is remembered from round to round)
 *       and the matches whose source is complete are pushed into a queue shared
 *       by the workgroup (one LDS atomic per wave and round).  After a barrier
 *       the queue is worked off DENSELY, entry t by thread t: the copy reads the
 *       source as aligned qwords, rotates it to the destination's alignment in
 *       registers (v_alignbyte) and writes aligned dwords; the ragged first and
 *       last dword go out as masked read-modify-writes inside the LDS
 *       (ds_mskor_b32), so neighbouring sequences never see a torn dword.  Then
 *       the copied bytes are marked valid and the next round begins.  A round
 *       resolves one level of the block's dependency DAG (about 19 levels of a
 *       hundred matches each on the C2 workload).  Overlapping matches
 *       (offset < length) run through the same engine: the period is doubled
 *       until it covers a copy unit, every step a non-overlapping copy.
 *   phase F (flush)     the window goes to the decoded slab with 16-byte
 *       coalesced stores (the window is placed so that LDS and HBM addresses
 *       are congruent modulo 16).
 *
 * Why this shape (tools/ubench_lds.hip, profiles/r02_ubench_lds.txt): an LDS access
 * that is not naturally aligned costs the CU one cycle per ACTIVE LANE (64 for a full
 * wave, whatever the width), an aligned one 4..9 cycles per wave instruction at random
 * addresses.  The previous generation of this kernel (git history: per-sequence done
 * bits, every lane polling for and copying its own match with unaligned 8-byte
 * accesses) spent about nine such lane-accesses per match, 18 k LDS cycles per
 * block.  Aligned accesses only pay when the lanes of an instruction are all busy:
 * hence the queue.  Literal stores stay unaligned dword stores on purpose: they are
 * sparse (few lanes per instruction), which is the one case the per-lane price wins.
 *
 * HBM traffic per block: payload once, sequence table once in (plus the entries
 * phase L looks up), decoded bytes once out.
 */
#include "la_dev.h"

#ifndef FAST_THREADS
#define FAST_THREADS 512
#endif
#ifndef LBATCH
#define LBATCH 2	/* payload chunks a thread has in flight in phase L */
#endif
#ifndef FAST_MIN_WAVES
#define FAST_MIN_WAVES 4
#endif
#define FAST_WAVES   (FAST_THREADS / 64)
#define FAST_QCAP    FAST_THREADS	/* queue entries per round: one per thread */
#define COPY_UNIT    40u		/* bytes one pass of the copy engine moves per lane */

/* Diagnostic build only (make diag, -DLA_DIAG): per-workgroup phase stamps go to a
 * buffer of their own; no output value depends on them. */
#ifdef LA_DIAG
__device__ unsigned long long *la_diagq_stamps;
#define STAMP(slot)                                                                       \
	do {                                                                              \
		if (threadIdx.x == 0 && la_diagq_stamps)                                    \
			la_diagq_stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
	} while (0)
#define STAMP_ADD(slot, v)                                                                \
	do {                                                                              \
		if (threadIdx.x == 0 && la_diagq_stamps)                                    \
			la_diagq_stamps[(size_t)blockIdx.x * 8 + (slot)] += (v);            \
	} while (0)
#else
#define STAMP(slot) do { } while (0)
#define STAMP_ADD(slot, v) do { } while (0)
#endif

/* LDS accepts unaligned 4-byte stores on gfx950 at one cycle per active lane (see above):
 * used for the sparse literal stores only. */
__device__ __forceinline__ void lds_st4(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

/* masked store of an aligned LDS dword, atomic inside the LDS: MEM = (MEM & ~mask) | data */
__device__ __forceinline__ void lds_mskor(uint32_t lds_addr, uint32_t mask, uint32_t data)
{
	asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(lds_addr), "v"(mask), "v"(data) : "memory");
}

/* A sequence-table entry travels in registers as one 64-bit value
 * (lit_src | lit_len << 16 | dst << 32 | off << 48): plain integers keep the
 * compiler from parking small structs in scratch memory. */
typedef uint64_t seq_t;
__device__ __forceinline__ seq_t seq_load(const la_lz4_seq *t, uint32_t k)
{
	return *(const uint64_t *)(const void *)(t + k);
}
#define SEQ_LIT_SRC(e) ((uint32_t)((e) & 0xFFFFu))
#define SEQ_LIT_LEN(e) ((uint32_t)(((e) >> 16) & 0xFFFFu))
#define SEQ_DST(e)     ((uint32_t)(((e) >> 32) & 0xFFFFu))
#define SEQ_OFF(e)     ((uint32_t)((e) >> 48))

/* set the bits [p, e) of the validity bitmap: three words straight-line (a copy unit or a usual
 * literal run), a wave-level loop for longer ranges (every lane of the wave must call) */
__device__ __forceinline__ void vbits_set(uint32_t *vb, uint32_t p, uint32_t e)
{
#define VBITS_STEP()                                                                          \
	if (p < e) {                                                                           \
		const uint32_t b = p & 31u;                                                    \
		uint32_t nb = 32u - b;                                                         \
		if (e - p < nb) nb = e - p;                                                    \
		const uint32_t m = (0xFFFFFFFFu >> (32u - nb)) << b;                           \
		atomicOr(&vb[p >> 5], m);                                                      \
		p += nb;                                                                       \
	}
	VBITS_STEP()
	VBITS_STEP()
	VBITS_STEP()
	while (__ballot(p < e) != 0) {
		VBITS_STEP()
	}
#undef VBITS_STEP
}

/*
 * The copy engine: window bytes [mdst, mdst + mlen) := window bytes [mdst - off, ...), LZ4 semantics
 * (off < mlen replicates with period off).  win16 = LDS offset arithmetic base: `wo` is the offset of
 * window byte 0 inside win[].  Every lane of the wave must call (wave-level loop); lanes without work
 * pass mlen = 0.  All LDS accesses are naturally aligned.
 */
__device__ __forceinline__ void match_copy(uint8_t *win, uint32_t wo, uint32_t mdst, uint32_t off, uint32_t mlen)
{
	const uint32_t lds_base = (uint32_t)(uintptr_t)win;	/* LDS byte address of win[0] (16-byte aligned) */
	uint32_t done = 0, eff = off;
	while (__ballot(done < mlen) != 0) {
		if (done < mlen) {
			/* eff = a multiple of the period that the bytes already in place allow
			 * (eff <= off + done): doubling it keeps every step a non-overlapping copy */
			if (eff < COPY_UNIT && 2u * eff <= off + done)
				eff *= 2u;
			uint32_t n = mlen - done;
			if (n > eff) n = eff;
			if (n > COPY_UNIT) n = COPY_UNIT;
			const uint32_t ad = wo + mdst + done;		/* destination, offset in win[] */
			const uint32_t as = ad - eff;			/* source */
			const uint32_t hb = ad & 3u;			/* bytes of the first destination dword in front of the copy */
			const uint32_t ad0 = ad - hb;
			const uint32_t nd = (hb + n + 3u) >> 2;		/* destination dwords touched: 1..11 */
			const uint32_t bs = as - hb;			/* source byte under byte 0 of the first destination dword */
			const uint32_t sh = bs & 3u;
			const uint32_t *const sp = (const uint32_t *)(const void *)(win + (bs - sh));	/* source dwords: sp[i], sp[i+1] make destination dword i */
			uint32_t S[11];
#pragma unroll
			for (uint32_t i = 0; i < 11; i++)
				S[i] = i <= nd ? sp[i] : 0u;	/* (the pair for the last dword is fetched on its own below) */
			const uint32_t t0 = sp[nd - 1u], t1 = sp[nd];
			/* first dword: bytes [hb, min(4, hb + n)) */
			const uint32_t he = hb + n < 4u ? hb + n : 4u;
			const uint32_t hmask = (0xFFFFFFFFu >> (32u - 8u * he)) & (0xFFFFFFFFu << (8u * hb));
			lds_mskor(lds_base + ad0, hmask, __builtin_amdgcn_alignbyte(S[1], S[0], sh) & hmask);
			/* whole dwords in between */
#pragma unroll
			for (uint32_t i = 1; i < 10; i++)
				if (i + 1u < nd)
					*(uint32_t *)(void *)(win + ad0 + 4u * i) = __builtin_amdgcn_alignbyte(S[i + 1], S[i], sh);
			if (nd > 1u) {
				const uint32_t te = ((hb + n - 1u) & 3u) + 1u;	/* bytes of the last dword */
				const uint32_t tmask = 0xFFFFFFFFu >> (32u - 8u * te);
				lds_mskor(lds_base + ad0 + 4u * (nd - 1u), tmask, __builtin_amdgcn_alignbyte(t1, t0, sh) & tmask);
			}
			done += n;
		}
	}
}

/* SEG = false: one workgroup per block of the table, blocks of at most MAXSEQ sequences (the
 * usual case).  SEG = true: the workgroups share out the few blocks with MORE sequences, listed
 * by lz4_classify_q_kernel, and run them in segments of MAXSEQ sequences. */
template <uint32_t MAXSEQ, bool SEG>
__global__ __launch_bounds__(FAST_THREADS, FAST_MIN_WAVES) void lz4_expand_queue_kernel(
    const uint8_t *__restrict__ src, uint64_t src_bytes, const la_lz4_block *__restrict__ blocks,
    uint32_t n, uint8_t *__restrict__ dst, uint64_t dst_cap, const uint64_t *__restrict__ dst_off,
    const uint32_t *__restrict__ out_len, uint32_t *status_out,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table,
    const uint64_t *__restrict__ table_off, const uint32_t *__restrict__ big_list,
    const uint32_t *__restrict__ big_count, uint32_t long_thr)
{
	const uint32_t *status = status_out;
	__shared__ __attribute__((aligned(16))) uint8_t win[16 + 65536 + 96];	/* 16 headroom (literal stores may start 3 bytes early) + 16 alignment shift + slack for over-reads */
	__shared__ __attribute__((aligned(16))) uint32_t vbits[2048 + 4];	/* one bit per window byte: the byte is final */
	/* phase L: per 32-byte payload chunk the first sequence with literals in or after it;
	 * phase M: the queue of matches whose source is complete (the phases are a barrier apart) */
	__shared__ __attribute__((aligned(16))) union {
		uint16_t chunk_first[2048 + 8];
		uint64_t queue[FAST_QCAP];
	} u;
	__shared__ uint32_t qtail[2], qmore[2];
	uint16_t *const chunk_first = u.chunk_first;

	auto do_block = [&](const uint32_t bi) __attribute__((always_inline)) {
	if (bi >= n)
		return;
	const la_lz4_block b = blocks[bi];
	const uint32_t olen = out_len[bi];
	const uint32_t ns_all = nseq[bi];
	const uint64_t doff = dst_off[bi];
	/* same predicate as the general kernel's skip test (0xFFFFFFFF: the block has no table) */
	if (status[bi] != LA_ST_OK || olen == 0 || !la_lz4_fast_eligible(b) || ns_all == 0xFFFFFFFFu ||
	    doff + olen > dst_cap || la_lz4_long_sequences(ns_all, olen, long_thr))
		return;		/* (few long sequences: the general kernel takes the block, la_dev.h) */

	STAMP(0);
	const uint32_t tid = threadIdx.x, lane = tid & 63;
	const la_lz4_seq *const tab_all = table + table_off[bi];
	const uint8_t *s = src + b.src_off;
	const uint64_t s_room = src_bytes - b.src_off;	/* bytes of the image from s on */
	uint8_t *g_out = dst + doff;
	const uint32_t wo = 16u + (uint32_t)((uintptr_t)g_out & 15);
	uint8_t *W = win + wo;	/* W[i] <-> g_out[i], congruent mod 16 */

	constexpr uint32_t MAXSTEPS = MAXSEQ / FAST_THREADS;
	/* the first two payload chunks of this thread (phase L) are requested right away:
	 * they depend on nothing, and their latency overlaps the table prepass */
	uint64_t vv[LBATCH][4];
#pragma unroll
	for (int uu = 0; uu < LBATCH; uu++) {
		const uint32_t c0 = (uu * FAST_THREADS + tid) << 5;
		vv[uu][0] = vv[uu][1] = vv[uu][2] = vv[uu][3] = 0;
		if (c0 < b.src_len) {
			if ((uint64_t)c0 + 32 <= s_room) {
				const uint4 a = ld_u128(s + c0), bq = ld_u128(s + c0 + 16);
				vv[uu][0] = ((uint64_t)a.y << 32) | a.x; vv[uu][1] = ((uint64_t)a.w << 32) | a.z;
				vv[uu][2] = ((uint64_t)bq.y << 32) | bq.x; vv[uu][3] = ((uint64_t)bq.w << 32) | bq.z;
			} else {
				/* last chunk of the image: never read past it */
				for (uint32_t i = 0; c0 + i < s_room && i < 32; i++) {
					const uint64_t by = (uint64_t)s[c0 + i] << (8 * (i & 7));
					if (i < 8) vv[uu][0] |= by; else if (i < 16) vv[uu][1] |= by;
					else if (i < 24) vv[uu][2] |= by; else vv[uu][3] |= by;
				}
			}
		}
	}
	/* the validity bitmap starts empty (16 bytes per thread) */
	*(uint4 *)(void *)(vbits + 4u * tid) = make_uint4(0, 0, 0, 0);
	if (tid < 4)
		vbits[2048 + tid] = 0;
	if (tid < 2) {
		qtail[tid] = 0;
		qmore[tid] = 0;
	}
	__syncthreads();

	/* Blocks with more sequences than a thread holds slots for are done in SEGMENTS of MAXSEQ
	 * sequences: prepass, literals and matches per segment, a barrier between segments.
	 * Sequences of earlier segments are complete by then (their bytes are marked valid).
	 * Inside a segment all sequence numbers are local (tab points at its first
	 * entry); payload chunks are numbered from cb, the chunk in which the segment's
	 * literals begin (it may be shared with the end of the previous segment: each side
	 * stores only its own sequences' bytes). */
	auto segment = [&](const uint32_t kb) __attribute__((always_inline)) {
	const la_lz4_seq *const tab = tab_all + kb;
	const uint32_t ns = ns_all - kb < MAXSEQ ? ns_all - kb : MAXSEQ;
	const uint32_t nslots = (ns + FAST_THREADS - 1u) / FAST_THREADS;
	uint32_t cb = 0, seg_prev_end = 0;
	if (kb) {
		const seq_t pvb = seq_load(tab_all, kb - 1);	/* same address in every thread */
		seg_prev_end = SEQ_LIT_SRC(pvb) + SEQ_LIT_LEN(pvb);
		cb = seg_prev_end >> 5;
	}
	/* This thread's sequences: k = r * FAST_THREADS + tid.  Per sequence the match phase keeps
	 *   mo[r]  = match destination | offset << 16
	 *   ml[r]  = match length
	 *   ar[r]  = bitmap word under the first source byte still to be seen valid | source bytes
	 *            beyond the 64 bits from that word on << 16 (0 for a source of up to 33 bytes)
	 *   m0[r], m1[r] = the source's bits inside those two words
	 * and one bit of `pend` while the match is not queued yet. */
	uint32_t mo[MAXSTEPS], ml[MAXSTEPS], ar[MAXSTEPS], m0[MAXSTEPS], m1[MAXSTEPS];
	uint32_t pend = 0;
	if (tid == 0 && (cb << 5) < seg_prev_end)
		chunk_first[0] = 0;	/* chunk shared with the previous segment: this side starts with its first sequence */
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		mo[r] = ml[r] = ar[r] = m0[r] = m1[r] = 0;
		if (r >= nslots)
			continue;
		const uint32_t k = r * FAST_THREADS + tid;
		const bool have = k < ns;
		const seq_t e = have ? seq_load(tab, k) : 0;
		const seq_t pv = (have && (k > 0 || kb > 0)) ? seq_load(tab_all, kb + k - 1) : 0;
		const uint32_t next_dst = (have && kb + k + 1 < ns_all) ? SEQ_DST(seq_load(tab_all, kb + k + 1)) : olen;
		const uint32_t prev_end = SEQ_LIT_SRC(pv) + SEQ_LIT_LEN(pv);
		const uint32_t d = SEQ_DST(e), ll = SEQ_LIT_LEN(e), off = SEQ_OFF(e);
		const uint32_t mdst = d + ll;
		const uint32_t mlen = have ? next_dst - mdst : 0u;
		if (have) {
			/* chunk_first[c] = first sequence whose literals end beyond payload offset 32c */
			const uint32_t le = SEQ_LIT_SRC(e) + ll;
			for (uint32_t c = (prev_end + 31) >> 5; (c << 5) < le; c++)
				chunk_first[c - cb] = (uint16_t)k;
			if (k + 1 == ns)
				chunk_first[2048] = (uint16_t)(((le + 31) >> 5) - cb);	/* chunks from here on hold no literals of this segment */
		}
		/* literal bytes count as valid from the start: phase L puts all of them in place
		 * before the first match looks */
		vbits_set(vbits, d, have ? mdst : d);
		if (have && mlen != 0 && off != 0 && off <= mdst) {
			const uint32_t span = mlen < off ? mlen : off;
			mo[r] = mdst | (off << 16);
			ml[r] = mlen;
			const uint32_t s0 = mdst - off, bo = s0 & 31u;
			const uint32_t fit = span < 64u - bo ? span : 64u - bo;
			const uint64_t mk = (0xFFFFFFFFFFFFFFFFull >> (64u - fit)) << bo;
			ar[r] = (s0 >> 5) | ((span - fit) << 16);
			m0[r] = (uint32_t)mk;
			m1[r] = (uint32_t)(mk >> 32);
			pend |= 1u << r;
		}
	}
	__syncthreads();

	/* ---- phase L: literals, one thread per 32-byte payload chunk ----
	 * Two coalesced 16-byte loads of the compressed stream per chunk; the chunk's
	 * sequences come from chunk_first (LDS) and six table entries fetched together;
	 * literal bytes go to the window as whole dwords.  Work is balanced
	 * by payload bytes, and every payload byte is read once. */
	const uint32_t nlit_chunks = chunk_first[2048];
	for (uint32_t base = 0; base < nlit_chunks; base += LBATCH * FAST_THREADS) {
		uint32_t kk[LBATCH];
		seq_t pe[LBATCH][6];
#pragma unroll
		for (int uu = 0; uu < LBATCH; uu++) {
			const uint32_t c = base + uu * FAST_THREADS + tid;	/* chunk number inside the segment */
			kk[uu] = c < nlit_chunks ? (uint32_t)chunk_first[c] : 0xFFFFFFFFu;
			if (base != 0 || kb != 0) {	/* payloads beyond 32 KiB, later segments: chunks are loaded here */
				vv[uu][0] = vv[uu][1] = vv[uu][2] = vv[uu][3] = 0;
				const uint32_t c0 = (cb + c) << 5;
				if (c < nlit_chunks) {
					if ((uint64_t)c0 + 32 <= s_room) {
						const uint4 a = ld_u128(s + c0), bq = ld_u128(s + c0 + 16);
						vv[uu][0] = ((uint64_t)a.y << 32) | a.x; vv[uu][1] = ((uint64_t)a.w << 32) | a.z;
						vv[uu][2] = ((uint64_t)bq.y << 32) | bq.x; vv[uu][3] = ((uint64_t)bq.w << 32) | bq.z;
					} else {
						for (uint32_t i = 0; c0 + i < s_room && i < 32; i++) {
							const uint64_t by = (uint64_t)s[c0 + i] << (8 * (i & 7));
							if (i < 8) vv[uu][0] |= by; else if (i < 16) vv[uu][1] |= by;
							else if (i < 24) vv[uu][2] |= by; else vv[uu][3] |= by;
						}
					}
				}
			}
		}
#pragma unroll
		for (int uu = 0; uu < LBATCH; uu++) {
#pragma unroll
			for (int t = 0; t < 6; t++)
				pe[uu][t] = (kk[uu] != 0xFFFFFFFFu && kk[uu] + t < ns) ? seq_load(tab, kk[uu] + t) : 0;
		}
#pragma unroll
		for (int uu = 0; uu < LBATCH; uu++) {
			if (kk[uu] == 0xFFFFFFFFu)
				continue;
			const uint32_t c0 = (cb + base + uu * FAST_THREADS + tid) << 5, c1 = c0 + 32;
			/* The chunk's eight dwords go out with STATIC register indices: dword m of
			 * the chunk belongs to at most one literal run (two runs are at least a
			 * 3-byte sequence header apart), and a whole-dword store may spill up to
			 * three bytes over either end of the run: those bytes lie in the match
			 * before / after the run (a match is at least four bytes long), which
			 * phase M writes after the barrier, exactly.  No shifting, no register selects. */
			const uint32_t dd[8] = { (uint32_t)vv[uu][0], (uint32_t)(vv[uu][0] >> 32), (uint32_t)vv[uu][1], (uint32_t)(vv[uu][1] >> 32),
			    (uint32_t)vv[uu][2], (uint32_t)(vv[uu][2] >> 32), (uint32_t)vv[uu][3], (uint32_t)(vv[uu][3] >> 32) };
			/* returns true when the chunk is finished.  exact_head: the run is the first of a
			 * later segment -- the match in front of it is FINAL (previous segment), so the
			 * first dword must not spill backwards: its bytes go out one by one. */
			auto put = [&](const seq_t e, const bool exact_head) -> bool {
				const uint32_t ls = SEQ_LIT_SRC(e), le = ls + SEQ_LIT_LEN(e);
				if (ls >= c1)
					return true;
				const uint32_t lo = ls > c0 ? ls : c0, hi = le < c1 ? le : c1;
				if (hi > lo) {
					uint32_t m0 = (lo - c0) >> 2;
					const uint32_t mend = (hi - c0 + 3) >> 2;
					uint8_t *wb = W + SEQ_DST(e) + c0 - ls;	/* wb[p - c0] <-> payload[p]; W has 16 bytes of headroom */
					if (exact_head && lo == ls && ((lo - c0) & 3u)) {
						const uint32_t stop = hi < c0 + 4 * m0 + 4 ? hi : c0 + 4 * m0 + 4;
						for (uint32_t pp = lo; pp < stop; pp++)	/* (once per segment: straight from the image) */
							wb[pp - c0] = (uint64_t)pp < s_room ? s[pp] : (uint8_t)0;
						m0++;
					}
					const uint32_t cnt = mend > m0 ? mend - m0 : 0;
#pragma unroll
					for (uint32_t m = 0; m < 8; m++)
						if (m - m0 < cnt)
							lds_st4(wb + 4 * m, dd[m]);
				}
				return le >= c1;
			};
			bool fin = false;
#pragma unroll
			for (int t = 0; t < 6; t++)
				if (!fin)
					fin = (kk[uu] + t >= ns) || put(pe[uu][t], kb != 0 && kk[uu] + t == 0);
			for (uint32_t k = kk[uu] + 6; !fin && k < ns; k++)	/* more than six sequences touch this chunk: rare */
				fin = put(seq_load(tab, k), false);
		}
	}
	STAMP(1);
	__syncthreads();
	STAMP(2);

	/* ---- phase M: matches, in rounds (see the header) ---- */
#ifdef LA_DIAG
	unsigned long long dg_scan = 0, dg_copy = 0, dg_bar = 0, dg_rounds = 0;
#define DG_T() __builtin_readcyclecounter()
#endif
	for (uint32_t round = 0;; round++) {
		const uint32_t par = round & 1u;
#ifdef LA_DIAG
		const unsigned long long dg0 = DG_T();
#endif
		/* look: which of this thread's matches have their whole source in place? */
		if (__ballot(pend != 0) != 0) {
			uint32_t pick = 0xFFu, d_lo = 0, d_hi = 0;
#pragma unroll
			for (uint32_t r = 0; r < MAXSTEPS; r++) {
				if (r >= nslots)
					continue;
				const bool p = (pend >> r) & 1u;
				if (__ballot(p) == 0)
					continue;
				bool pass;
				for (;;) {
					const uint32_t *const q = vbits + (ar[r] & 0xFFFFu);
					const uint32_t w0 = q[0], w1 = q[1];
					pass = p && (w0 & m0[r]) == m0[r] && (w1 & m1[r]) == m1[r];
					/* a source longer than its 64 bits (rare): on to the next two words */
					const bool again = pass && (ar[r] >> 16) != 0;
					if (__ballot(again) == 0)
						break;
					if (again) {
						const uint32_t rest = ar[r] >> 16;
						const uint32_t fit = rest < 64u ? rest : 64u;
						const uint64_t mk = 0xFFFFFFFFFFFFFFFFull >> (64u - fit);
						ar[r] = ((ar[r] & 0xFFFFu) + 2u) | ((rest - fit) << 16);
						m0[r] = (uint32_t)mk;
						m1[r] = (uint32_t)(mk >> 32);
					}
				}
				const bool rdy = pass;
				if (rdy && pick == 0xFFu) {
					pick = r;
					d_lo = mo[r];
					d_hi = ml[r];
				}
			}
			const bool has = pick != 0xFFu;
			const uint64_t bal = __ballot(has);
			if (bal != 0) {
				uint32_t qb = 0;
				if (lane == 0)
					qb = atomicAdd(&qtail[par], (uint32_t)__builtin_popcountll(bal));
				qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
				if (has && qb + rank < FAST_QCAP) {
					u.queue[qb + rank] = (uint64_t)d_lo | ((uint64_t)d_hi << 32);
					pend &= ~(1u << pick);
				}
			}
			if (__ballot(pend != 0) != 0 && lane == 0)
				qmore[par] = 1u;
		}
#ifdef LA_DIAG
		const unsigned long long dg1 = DG_T();
#endif
		__syncthreads();
#ifdef LA_DIAG
		const unsigned long long dg2 = DG_T();
		dg_scan += dg1 - dg0; dg_bar += dg2 - dg1; dg_rounds++;
#endif
		uint32_t qn = qtail[par];
		const uint32_t more = qmore[par];
		if (qn > FAST_QCAP) qn = FAST_QCAP;
		if (tid == 0) {
			qtail[par ^ 1u] = 0;
			qmore[par ^ 1u] = 0;
		}
		if (qn == 0) {
			/* nothing became ready: either everything is done, or the table is not one the
			 * parse kernel produced (a source that never completes) -- fail the block
			 * instead of spinning */
			if (more != 0 && tid == 0)
				status_out[bi] = LA_ST_LZ4_DECODE;
			break;
		}
		/* work the queue off, entry t by thread t; whole waves without an entry go straight on */
		if ((tid & ~63u) < qn) {
			uint32_t mdst = 0, off = 1, mlen = 0;
			if (tid < qn) {
				const uint64_t dsc = u.queue[tid];
				mdst = (uint32_t)dsc & 0xFFFFu;
				off = ((uint32_t)dsc >> 16) & 0xFFFFu;
				mlen = (uint32_t)(dsc >> 32);
			}
			match_copy(win, wo, mdst, off, mlen);
			vbits_set(vbits, mdst, mdst + mlen);
		}
#ifdef LA_DIAG
		const unsigned long long dg3 = DG_T();
#endif
		__syncthreads();
#ifdef LA_DIAG
		dg_copy += dg3 - dg2; dg_bar += DG_T() - dg3;
#endif
	}
#ifdef LA_DIAG
	if (threadIdx.x == 0 && la_diagq_stamps) {
		la_diagq_stamps[(size_t)blockIdx.x * 8 + 6] = dg_scan | (dg_rounds << 40);
		la_diagq_stamps[(size_t)blockIdx.x * 8 + 7] = dg_copy | ((dg_bar >> 4) << 40);
	}
#endif
	STAMP(3);
	__syncthreads();
	};
	/* the usual block is one segment: that copy of the body is compiled with kb = 0 folded in
	 * (and without the register pressure of a loop around it) */
	if (!SEG) {
		if (ns_all > MAXSEQ)
			return;		/* the SEG launch takes it */
		segment(0u);
	} else {
		if (ns_all <= MAXSEQ)
			return;
		for (uint32_t kb = 0; kb < ns_all; kb += MAXSEQ)
			segment(kb);
	}
	STAMP(4);

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
	STAMP(5);
	};	/* do_block */
	if (!SEG) {
		do_block(blockIdx.x);
	} else {
		const uint32_t cnt = *big_count;
		for (uint32_t li = blockIdx.x; li < cnt; li += gridDim.x) {
			do_block(big_list[li]);
			__syncthreads();	/* the window is reused */
		}
	}
}

#ifdef LA_DIAG
extern "C" int la_diagq_set_stamps(void *d_buf)
{
	unsigned long long *p = (unsigned long long *)d_buf;
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(la_diagq_stamps), &p, sizeof(p));
}
#endif

/* blocks the SEG launch must take: eligible, with a table, more than MAXSEQ sequences */
__global__ __launch_bounds__(256) void lz4_classify_q_kernel(const la_lz4_block *__restrict__ blocks, uint32_t n,
    const uint32_t *__restrict__ status, const uint32_t *__restrict__ nseq, uint32_t *__restrict__ big_list,
    uint32_t *__restrict__ big_count, const uint32_t *__restrict__ out_len, uint32_t long_thr)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t ns = nseq[i];
	if (ns != 0xFFFFFFFFu && ns > LA_LZ4_FAST_MAXSEQ && status[i] == LA_ST_OK && la_lz4_fast_eligible(blocks[i]) &&
	    !la_lz4_long_sequences(ns, out_len[i], long_thr))
		big_list[atomicAdd(big_count, 1u)] = i;
}


void la_launch_lz4_expand_queue(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t long_thr)
{
	if (n == 0) return;
	hipLaunchKernelGGL((lz4_expand_queue_kernel<LA_LZ4_FAST_MAXSEQ, false>), dim3(n), dim3(FAST_THREADS), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq,
	    d_table, d_table_off, (const uint32_t *)nullptr, (const uint32_t *)nullptr, long_thr);
}

/* blocks with more than LA_LZ4_FAST_MAXSEQ sequences, over the WHOLE table: classify + a small
 * grid that shares them out.  d_big: n + 1 words of workspace (count first). */
void la_launch_lz4_expand_queue_big(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t *d_big, uint32_t long_thr)
{
	if (n == 0) return;
	(void)hipMemsetAsync(d_big, 0, sizeof(uint32_t), s);
	hipLaunchKernelGGL(lz4_classify_q_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_blocks, n, d_status, d_nseq,
	    d_big + 1, d_big, d_out_len, long_thr);
	const uint32_t grid = n < 1024u ? n : 1024u;
	hipLaunchKernelGGL((lz4_expand_queue_kernel<LA_LZ4_FAST_MAXSEQ, true>), dim3(grid), dim3(FAST_THREADS), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq,
	    d_table, d_table_off, (const uint32_t *)(d_big + 1), (const uint32_t *)d_big, long_thr);
}
