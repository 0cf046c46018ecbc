/*
 * la_lz4_fast.hip -- LDS-window LZ4 expand kernel (gfx950): the hot kernel of
 * the lz4 filter path for independent blocks of at most 64 KiB
 * (libarchive/archive_read_support_filter_lz4.c:557-561, the
 * LZ4_decompress_safe call of BD=4 frames).
 *
 * One 512-thread workgroup owns one block; its whole output lives in a 64 KiB
 * LDS window (two workgroups per CU: 2 x 77 KiB of the 160 KiB LDS).  The
 * parse kernel has already reduced the token chain to a table of sequences
 * {literal source, literal length, output position, match offset}, so inside
 * the block every copy is known up front:
 *
 *   prepass             every thread loads its sequences' entries (kept in
 *       registers for all phases), publishes their output positions in LDS and
 *       builds the chunk index: for every 32-byte chunk of the payload, the
 *       first sequence that still has literals in or after it.
 *   phase L (literals)  one thread per 32-byte payload chunk: two coalesced
 *       16-byte loads of the compressed stream, the chunk's sequences from the
 *       chunk index, literal bytes into the window as whole dwords with static
 *       register indices.  Work is balanced by bytes, not by sequences; every
 *       payload byte is read once.
 *   phase M (matches)   one thread per sequence, sequences taken in increasing
 *       order.  A match may read bytes an earlier match produces, so every
 *       sequence publishes a "done" bit in LDS and a match waits only for the
 *       (typically one or two) earlier MATCHES that overlap its source range,
 *       found by a binary search over the output positions.  Dependencies
 *       always point to lower sequence numbers and waves take sequences in
 *       increasing order, so the lowest unfinished sequence can always run: no
 *       deadlock, no barrier inside the phase; every spin is bounded anyway.
 *   phase F (flush)     the window goes to the decoded slab with 16-byte
 *       coalesced stores (the window is placed so that LDS and HBM addresses
 *       are congruent modulo 16).
 *
 * HBM traffic per block: payload once, sequence table once in (plus the entries
 * phase L looks up), decoded bytes once out.
 *
 * What bounds it (PMC, C2 workload): the LDS pipe is 58 % busy, the SIMDs a little
 * under half, and behind both sits the dependency chain of phase M (DAG depth 16 per
 * block, about 31 poll iterations per wave).  What a poll iteration costs is LDS
 * round trips in series, LDS instructions, and VALU instructions, in that order.
 * Measured on the way here (16 GiB C2 stream, expand ms):
 *   - all loads of a match copy before any store, ragged end as an overlapping
 *     end-aligned store                                           39.1 -> 32.8
 *   - literal chunks as whole dwords, static register indices     32.7 -> 27.8
 *   - no wait for a sequence touched only in its literals          27.8 -> 26.9
 *   - short (4..7 byte) matches in the same load batch            26.9 -> 24.8
 *   - one look at the flag word per iteration                      24.8 -> 23.5
 *   - round 2: 16-byte unaligned accesses for matches of 16 bytes and more, two
 *     8-byte ones below, no load a lane does not need (an unaligned LDS access
 *     costs one cycle per active lane whatever its width)         23.2 -> 21.7
 * and slower, so not kept: neighbouring literal dwords paired into 8-byte stores
 * (22.3 from 21.7: the extra predicates cost more than the lane-stores saved),
 * skipping the register slots beyond the block's last sequence in the prepass and
 * the search (no change: those phases wait for memory, not for issue slots), a match
 * phase in two passes (every slot gets one to three looks, the stragglers of all slots
 * are taken in a second pass: 23.3 / 23.5 / 23.8 from 21.8 -- the lanes a slot waits for
 * are not a few stragglers, the whole in-flight set advances one dependency level per look),
 * pulling the table and payload of the block 512 / 1024 / 2048 places ahead towards L2 at the
 * start of the match phase (22.4 / 22.5 / 22.8 from 21.8: the prepass is not waiting for HBM),
 * the source loads of a match issued right behind its flag look instead of after it (safe: the LDS
 * serves a wave in order; one round trip per level instead of two, but 24.3 from 21.8 -- the loads of
 * lanes that are not ready yet are lane-accesses the LDS pipe has no room for),
 * wave-cooperative match copies (four matches per pass
 * through ds_bpermute, 35.1), flag look requested one iteration ahead (28.3),
 * speculative source read behind the flag look (27.6), 8-ary search (26.9 from
 * 23.5: more LDS instructions), per-sequence literal copies from global memory
 * (43.6), 256- and 1024-thread workgroups (no gain), longer poll sleeps (+1..4 %), every
 * lane walking through its own sequences at its own pace instead of the wave finishing
 * slot r first (28.0 from 23.2: the per-lane register picks cost more than the waits);
 * and without effect: 1, 2, 8 or 16 consecutive sequences per lane group instead of 4,
 * the early payload loads moved behind the table loads / made branch-free (the compiler
 * waits for them right behind the loads because a byte-wise tail path defines the same
 * registers; removing that wait changed nothing measurable).
 */
#include "la_dev.h"

#ifndef POLL_SLEEP
#define POLL_SLEEP 1
#endif
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

/* Diagnostic build only (make diag, -DLA_DIAG): per-workgroup phase stamps go to a
 * buffer of their own; no output value depends on them. */
#ifdef LA_DIAG
__device__ unsigned long long *la_diag_stamps;
#define STAMP(slot)                                                                       \
	do {                                                                              \
		if (threadIdx.x == 0 && la_diag_stamps)                                    \
			la_diag_stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
	} while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

/* A sequence-table entry travels in registers as one 64-bit value
 * (lit_src | lit_len << 16 | dst << 32 | off << 48): plain integers keep the
 * compiler from parking small structs in scratch memory. */
typedef uint64_t seq_t;

/*
 * Ownership of sequences in the match phase: a STEP is sixteen consecutive sequences
 * (16g .. 16g+15); step g belongs to wave g % 8, which runs its steps in increasing
 * order.  Register slot r of lane l in wave w holds sequence
 *     k = (((4 r + l/16) * 8 + w) * 16) + l % 16
 * so that the step a wave executes at (r, t) sits in lanes 16t .. 16t+15.
 */
#ifndef STEP_SHIFT
#define STEP_SHIFT  2u		/* log2(sequences per step) */
#endif
#define STEP_SEQS   (1u << STEP_SHIFT)		/* sequences per step */
#define STEP_LANES  (64u >> STEP_SHIFT)		/* lanes per sequence, 8 bytes each per pass */
#define STEPS_PER_SLOT (64u >> STEP_SHIFT)	/* steps held by one register slot of a wave */
__device__ __forceinline__ uint32_t fast_seq_index(uint32_t r, uint32_t wave, uint32_t lane)
{
	return (((r * STEPS_PER_SLOT + (lane >> STEP_SHIFT)) * FAST_WAVES + wave) << STEP_SHIFT) + (lane & (STEP_SEQS - 1));
}
__device__ __forceinline__ seq_t seq_load(const la_lz4_seq *t, uint32_t k)
{
	return *(const uint64_t *)(const void *)(t + k);
}
#define SEQ_LIT_SRC(e) ((uint32_t)((e) & 0xFFFFu))
#define SEQ_LIT_LEN(e) ((uint32_t)(((e) >> 16) & 0xFFFFu))
#define SEQ_DST(e)     ((uint32_t)(((e) >> 32) & 0xFFFFu))
#define SEQ_OFF(e)     ((uint32_t)((e) >> 48))

/* SEG = false: one workgroup per block of the table, blocks of at most MAXSEQ sequences (the
 * usual case).  SEG = true: the workgroups share out the few blocks with MORE sequences, listed
 * by lz4_classify_kernel, and run them in segments of MAXSEQ sequences. */
template <uint32_t MAXSEQ, bool SEG>
__global__ __launch_bounds__(FAST_THREADS, FAST_MIN_WAVES) void lz4_expand_fast_kernel(
    const uint8_t *__restrict__ src, uint64_t src_bytes, const la_lz4_block *__restrict__ blocks,
    uint32_t n, uint8_t *__restrict__ dst, uint64_t dst_cap, const uint64_t *__restrict__ dst_off,
    const uint32_t *__restrict__ out_len, uint32_t *status_out,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table,
    const uint64_t *__restrict__ table_off, const uint32_t *__restrict__ big_list,
    const uint32_t *__restrict__ big_count, uint32_t long_thr)
{
	const uint32_t *status = status_out;
	__shared__ __attribute__((aligned(16))) uint8_t win[16 + 65536 + 96];	/* 16 headroom (literal stores may start 3 bytes early) + 16 alignment shift + slack for over-reads */
	__shared__ uint16_t dstpos[MAXSEQ + 4];
	__shared__ uint32_t donebits[MAXSEQ / 32];	/* one bit per sequence: its match is in the window */
	__shared__ uint16_t chunk_first[2048 + 8];	/* per 32-byte payload chunk: first sequence with literals in or after it */

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
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const la_lz4_seq *const tab_all = table + table_off[bi];
	const uint8_t *s = src + b.src_off;
	const uint64_t s_room = src_bytes - b.src_off;	/* bytes of the image from s on */
	uint8_t *g_out = dst + doff;
	uint8_t *W = win + 16 + ((uintptr_t)g_out & 15);	/* W[i] <-> g_out[i], congruent mod 16 */

	/* This thread's sequences (fast_seq_index): their table entries are fetched once, up
	 * front, and stay in registers for every pass over them.  The entry of the sequence
	 * before each one comes along: its literal end tells which payload chunks START their
	 * literals with this sequence. */
	constexpr uint32_t MAXSTEPS = MAXSEQ / FAST_THREADS;
	/* the first two payload chunks of this thread (phase L) are requested right away:
	 * they depend on nothing, and their latency overlaps the table prepass */
	uint64_t vv[LBATCH][4];
#pragma unroll
	for (int u = 0; u < LBATCH; u++) {
		const uint32_t c0 = (u * FAST_THREADS + tid) << 5;
		vv[u][0] = vv[u][1] = vv[u][2] = vv[u][3] = 0;
		if (c0 < b.src_len) {
			if ((uint64_t)c0 + 32 <= s_room) {
				const uint4 a = ld_u128(s + c0), bq = ld_u128(s + c0 + 16);
				vv[u][0] = ((uint64_t)a.y << 32) | a.x; vv[u][1] = ((uint64_t)a.w << 32) | a.z;
				vv[u][2] = ((uint64_t)bq.y << 32) | bq.x; vv[u][3] = ((uint64_t)bq.w << 32) | bq.z;
			} else {
				/* last chunk of the image: never read past it */
				for (uint32_t i = 0; c0 + i < s_room && i < 32; i++) {
					const uint64_t by = (uint64_t)s[c0 + i] << (8 * (i & 7));
					if (i < 8) vv[u][0] |= by; else if (i < 16) vv[u][1] |= by;
					else if (i < 24) vv[u][2] |= by; else vv[u][3] |= by;
				}
			}
		}
	}
	/* Blocks with more sequences than the LDS arrays hold are done in SEGMENTS of MAXSEQ
	 * sequences: prepass, literals and matches per segment, a barrier between segments.
	 * Sequences of earlier segments are complete by then, so a segment only ever waits on
	 * its own.  Inside a segment all sequence numbers are local (tab points at its first
	 * entry); payload chunks are numbered from cb, the chunk in which the segment's
	 * literals begin (it may be shared with the end of the previous segment: each side
	 * stores only its own sequences' bytes). */
	auto segment = [&](const uint32_t kb) __attribute__((always_inline)) {
	const la_lz4_seq *const tab = tab_all + kb;
	const uint32_t ns = ns_all - kb < MAXSEQ ? ns_all - kb : MAXSEQ;
	uint32_t cb = 0, seg_prev_end = 0;
	if (kb) {
		const seq_t pvb = seq_load(tab_all, kb - 1);	/* same address in every thread */
		seg_prev_end = SEQ_LIT_SRC(pvb) + SEQ_LIT_LEN(pvb);
		cb = seg_prev_end >> 5;
	}
	seq_t ent[MAXSTEPS];
	uint32_t prev_end[MAXSTEPS];
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		ent[r] = 0;
		prev_end[r] = 0;
	}
	/* (slot r holds sequences r * FAST_THREADS and up: slots beyond the segment's last sequence
	 * are skipped wave-uniformly in every pass below -- the kernel is bound by instruction issue) */
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		if (r * FAST_THREADS >= ns)
			break;
		const uint32_t k = fast_seq_index(r, wave, lane);
		ent[r] = k < ns ? seq_load(tab, k) : 0;
		const seq_t pv = ((k > 0 || kb > 0) && k < ns) ? seq_load(tab_all, kb + k - 1) : 0;
		prev_end[r] = SEQ_LIT_SRC(pv) + SEQ_LIT_LEN(pv);
	}
	if (tid < MAXSEQ / 32)
		donebits[tid] = 0;
	if (tid == 0) {
		/* output position where the segment ends: the match length of its last sequence */
		dstpos[ns] = kb + ns < ns_all ? (uint16_t)SEQ_DST(seq_load(tab_all, kb + ns)) : (uint16_t)olen;
		if ((cb << 5) < seg_prev_end)
			chunk_first[0] = 0;	/* chunk shared with the previous segment: this side starts with its first sequence */
	}
	/* chunk_first[c] = first sequence whose literals end beyond payload offset 32c */
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		if (r * FAST_THREADS >= ns)
			break;
		const uint32_t k = fast_seq_index(r, wave, lane);
		if (k < ns) {
			const uint32_t le = SEQ_LIT_SRC(ent[r]) + SEQ_LIT_LEN(ent[r]);
			for (uint32_t c = (prev_end[r] + 31) >> 5; (c << 5) < le; c++)
				chunk_first[c - cb] = (uint16_t)k;
			dstpos[k] = (uint16_t)SEQ_DST(ent[r]);
			if (k + 1 == ns)
				chunk_first[2048] = (uint16_t)(((le + 31) >> 5) - cb);	/* chunks from here on hold no literals of this segment */
		}
	}
	__syncthreads();
#ifdef LA_DIAG_L
	STAMP(6);
#endif

	/* ---- phase L: literals, one thread per 32-byte payload chunk ----
	 * Two coalesced 16-byte loads of the compressed stream per chunk; the chunk's
	 * sequences come from chunk_first (LDS) and six table entries fetched together;
	 * literal bytes go to the window with unaligned 8-byte LDS stores.  Work is balanced
	 * by payload bytes, and every payload byte is read once. */
	const uint32_t nlit_chunks = chunk_first[2048];
	for (uint32_t base = 0; base < nlit_chunks; base += LBATCH * FAST_THREADS) {
		uint32_t kk[LBATCH];
		seq_t pe[LBATCH][6];
#pragma unroll
		for (int u = 0; u < LBATCH; u++) {
			const uint32_t c = base + u * FAST_THREADS + tid;	/* chunk number inside the segment */
			kk[u] = c < nlit_chunks ? (uint32_t)chunk_first[c] : 0xFFFFFFFFu;
			if (base != 0 || kb != 0) {	/* payloads beyond 32 KiB, later segments: chunks are loaded here */
				vv[u][0] = vv[u][1] = vv[u][2] = vv[u][3] = 0;
				const uint32_t c0 = (cb + c) << 5;
				if (c < nlit_chunks) {
					if ((uint64_t)c0 + 32 <= s_room) {
						const uint4 a = ld_u128(s + c0), bq = ld_u128(s + c0 + 16);
						vv[u][0] = ((uint64_t)a.y << 32) | a.x; vv[u][1] = ((uint64_t)a.w << 32) | a.z;
						vv[u][2] = ((uint64_t)bq.y << 32) | bq.x; vv[u][3] = ((uint64_t)bq.w << 32) | bq.z;
					} else {
						for (uint32_t i = 0; c0 + i < s_room && i < 32; i++) {
							const uint64_t by = (uint64_t)s[c0 + i] << (8 * (i & 7));
							if (i < 8) vv[u][0] |= by; else if (i < 16) vv[u][1] |= by;
							else if (i < 24) vv[u][2] |= by; else vv[u][3] |= by;
						}
					}
				}
			}
		}
#pragma unroll
		for (int u = 0; u < LBATCH; u++) {
#pragma unroll
			for (int t = 0; t < 6; t++)
				pe[u][t] = (kk[u] != 0xFFFFFFFFu && kk[u] + t < ns) ? seq_load(tab, kk[u] + t) : 0;
		}
#ifdef LA_DIAG_L
		if (base == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(7); }
#endif
#pragma unroll
		for (int u = 0; u < LBATCH; u++) {
			if (kk[u] == 0xFFFFFFFFu)
				continue;
			const uint32_t c0 = (cb + base + u * FAST_THREADS + tid) << 5, c1 = c0 + 32;
			/* The chunk's eight dwords go out with STATIC register indices: dword m of
			 * the chunk belongs to at most one literal run (two runs are at least a
			 * 3-byte sequence header apart), and a whole-dword store may spill up to
			 * three bytes over either end of the run: those bytes lie in the match
			 * before / after the run (a match is at least four bytes long), which
			 * phase M writes after the barrier.  No shifting, no register selects. */
			const uint32_t dd[8] = { (uint32_t)vv[u][0], (uint32_t)(vv[u][0] >> 32), (uint32_t)vv[u][1], (uint32_t)(vv[u][1] >> 32),
			    (uint32_t)vv[u][2], (uint32_t)(vv[u][2] >> 32), (uint32_t)vv[u][3], (uint32_t)(vv[u][3] >> 32) };
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
					fin = (kk[u] + t >= ns) || put(pe[u][t], kb != 0 && kk[u] + t == 0);
			for (uint32_t k = kk[u] + 6; !fin && k < ns; k++)	/* more than six sequences touch this chunk: rare */
				fin = put(seq_load(tab, k), false);
		}
	}
	STAMP(1);
	__syncthreads();
	STAMP(2);

	/* ---- phase M: matches, one thread per sequence ----
	 * First every step's dependency range [q, qend] (branch-free binary searches over
	 * the output positions, interleaved across steps), then the steps in order. */
	uint32_t qcur[MAXSTEPS];	/* search cursor, then (first dependency | count << 16) */
#ifndef FAST_BSEARCH
	/* The sequence under a source position comes from a MAP, not from a binary search over the output
	 * positions (twelve dependent LDS reads per sequence: a third of the kernel's LDS cycles, PMC + count
	 * in DESIGN.md 5c): for every 32-byte chunk of the output the sequence that holds the chunk's first
	 * byte, then a short walk forward (a chunk holds at most eight sequence starts; half a one on the C2
	 * stream).  The map lives where phase L kept its chunk index: that is dead behind the barrier above. */
	uint16_t *const cmap = chunk_first;
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		if (r * FAST_THREADS >= ns)
			break;
		const uint32_t k = fast_seq_index(r, wave, lane);
		if (k < ns) {
			const uint32_t d = SEQ_DST(ent[r]);
			uint32_t nxt = dstpos[k + 1];
			if (nxt < d)
				nxt = 65536u;	/* (positions are 16-bit: an end at 65536 reads 0; a lone sequence that
						 * fills the window leaves the map unset, and nobody asks) */
			if (k == 0)
				for (uint32_t c = 0; (c << 5) < d; c++)
					cmap[c] = 0;	/* (later segments: positions in front of the segment) */
			for (uint32_t c = (d + 31) >> 5; (c << 5) < nxt; c++)
				cmap[c] = (uint16_t)k;
		}
	}
	__syncthreads();
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		qcur[r] = 0;
		if (r * FAST_THREADS >= ns)
			continue;
		const uint32_t k = fast_seq_index(r, wave, lane);
		const uint32_t s0 = SEQ_DST(ent[r]) + SEQ_LIT_LEN(ent[r]) - SEQ_OFF(ent[r]);
		if (k > 0 && k < ns) {
			const uint32_t q0 = cmap[(s0 < 65535u ? s0 : 65535u) >> 5];
			qcur[r] = q0 < k - 1 ? q0 : k - 1;	/* largest idx < k with dstpos[idx] <= s0, from below */
		}
	}
	{
		uint32_t act = 0;	/* wave-uniform: slots in which some lane is still walking */
#pragma unroll
		for (uint32_t r = 0; r < MAXSTEPS; r++)
			if (r * FAST_THREADS < ns)
				act |= 1u << r;
		while (act) {
#pragma unroll
			for (uint32_t r = 0; r < MAXSTEPS; r++) {
				if (!((act >> r) & 1u))
					continue;
				const uint32_t k = fast_seq_index(r, wave, lane);
				const uint32_t s0 = SEQ_DST(ent[r]) + SEQ_LIT_LEN(ent[r]) - SEQ_OFF(ent[r]);
				const bool adv = k < ns && qcur[r] + 1 < k && dstpos[qcur[r] + 1] <= s0;
				if (adv)
					qcur[r]++;
				if (__ballot(adv) == 0)
					act &= ~(1u << r);
			}
		}
	}
#else
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++)
		qcur[r] = 0;
	for (uint32_t bit = MAXSEQ / 2; bit; bit >>= 1) {
#pragma unroll
		for (uint32_t r = 0; r < MAXSTEPS; r++) {
			if (r * FAST_THREADS >= ns)
				break;
			/* largest idx < k with dstpos[idx] <= s0 */
			const uint32_t k = fast_seq_index(r, wave, lane);
			const uint32_t s0 = SEQ_DST(ent[r]) + SEQ_LIT_LEN(ent[r]) - SEQ_OFF(ent[r]);
			const uint32_t cand = qcur[r] + bit;
			if (cand < k && k < ns && dstpos[cand] <= s0)
				qcur[r] = cand;
		}
	}
#endif
	/* Literals are all in the window already (phase L), so a match only has to wait for
	 * earlier MATCHES under its source range.  The last sequence the range touches is
	 * often touched in its literal part only (a sequence is literals first, match second):
	 * its entry tells, and then it drops out of the wait -- about a third of all waits,
	 * and many sequences end up waiting for nothing at all. */
	uint32_t qlh[MAXSTEPS];	/* last sequence under the source range | last source byte << 16; ~0: none */
#pragma unroll
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		qlh[r] = 0xFFFFFFFFu;
		if (r * FAST_THREADS >= ns)
			continue;
		const uint32_t k = fast_seq_index(r, wave, lane);
		const uint32_t d = SEQ_DST(ent[r]), mdst = d + SEQ_LIT_LEN(ent[r]), off = SEQ_OFF(ent[r]);
		const uint32_t s0 = mdst - off;
		const uint32_t next = k < ns ? dstpos[k + 1] : 0;	/* dstpos[ns] = where the segment's output ends */
		const uint32_t mlen = k < ns ? (next - mdst) & 0xFFFFu : 0;	/* (positions are 16-bit: an end at 65536 reads 0) */
		uint32_t cnt = 0;	/* number of earlier sequences (of this segment) to wait for */
		qlh[r] = 0xFFFFFFFFu;
		if (mlen != 0 && s0 < d) {
			const uint32_t span = mlen < off ? mlen : off;
			uint32_t hi_byte = s0 + span - 1;
			if (hi_byte >= d)
				hi_byte = d - 1;
			/* sources that end before this segment's first output byte need nothing here:
			 * earlier segments are complete */
			if (k > 0 && dstpos[0] <= hi_byte) {
				uint32_t qe = qcur[r];
				while (qe + 1 < k && dstpos[qe + 1] <= hi_byte)
					qe++;
				cnt = qe - qcur[r] + 1;
				qlh[r] = qe | (hi_byte << 16);
			}
		}
		qcur[r] = (qcur[r] & 0xFFFFu) | (cnt << 16);
	}
#ifdef LA_DIAG_M
	STAMP(6);	/* chunk map, walks and dependency ranges done */
#endif
#pragma unroll
	for (uint32_t h = 0; h < MAXSTEPS; h += 4) {	/* four entries in flight at a time: registers */
		if (h * FAST_THREADS >= ns)
			break;
		seq_t le[4];
#pragma unroll
		for (uint32_t r = 0; r < 4; r++)
			le[r] = qlh[h + r] != 0xFFFFFFFFu ? seq_load(tab, qlh[h + r] & 0xFFFFu) : 0;
#pragma unroll
		for (uint32_t r = 0; r < 4; r++)
			if (qlh[h + r] != 0xFFFFFFFFu && (qlh[h + r] >> 16) < SEQ_DST(le[r]) + SEQ_LIT_LEN(le[r]))
				qcur[h + r] -= 1u << 16;	/* source ends inside the literals of the last sequence: its match is not needed */
	}

#ifdef LA_DIAG_M
	STAMP(7);	/* the last touched sequence's entry looked up (global memory) */
#endif
#pragma unroll 1
	for (uint32_t r = 0; r < MAXSTEPS; r++) {
		if (r * FAST_THREADS >= ns)
			break;
		const uint32_t k = fast_seq_index(r, wave, lane);
		const bool active = k < ns;
		/* registers of step r (the array is indexed by a loop counter: pick by selects) */
		seq_t e = ent[0];
		uint32_t qp = qcur[0];
#pragma unroll
		for (uint32_t t = 1; t < MAXSTEPS; t++)
			if (r == t) { e = ent[t]; qp = qcur[t]; }
		const uint32_t off = SEQ_OFF(e);
		const uint32_t mdst = SEQ_DST(e) + SEQ_LIT_LEN(e);
		const uint32_t next = active ? dstpos[k + 1] : 0;
		const uint32_t mlen = active ? (next - mdst) & 0xFFFFu : 0;
		const uint32_t s0 = mdst - off;
		uint32_t q = qp & 0xFFFFu;
#ifdef FAST_EXP_NODEP
		const uint32_t qstop = q;	/* what-if (wrong output): nobody waits for anybody */
#else
		const uint32_t qstop = q + (qp >> 16);	/* exclusive; q == qstop: nothing to wait for */
#endif
		uint32_t spins = 0;
		bool fin = mlen == 0;
		if (active && fin)
			atomicOr(&donebits[k >> 5], 1u << (k & 31));

		for (;;) {
#if defined(LA_DIAG) && !defined(LA_DIAG_L) && !defined(LA_DIAG_M)
			if (lane == 0 && la_diag_stamps) atomicAdd(&la_diag_stamps[(size_t)blockIdx.x * 8 + 6], 1ull);
			const unsigned long long t_it0 = __builtin_readcyclecounter();
#endif
			if (!fin) {
				/* advance q over finished sequences: ONE look at the flag word of q per
				 * iteration (a second look in the same iteration would only be another LDS
				 * round trip for the whole wave; a wait that spans two words takes two
				 * iterations) */
				if (q < qstop) {
					/* bit 0 of `word` is the flag of q; the zeros shifted in from the
					 * top end the run at the word boundary */
					const uint32_t word = __hip_atomic_load(&donebits[q >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (q & 31);
					const uint32_t inv = ~word;
					q += inv ? (uint32_t)__builtin_ctz(inv) : 32u;
				}
				/* every wait is bounded: a dependency that never completes (impossible
				 * for a table the parse kernel produced) fails the block instead of
				 * hanging the GPU */
				if (q < qstop && ++spins > (1u << 20)) {
					status_out[bi] = LA_ST_LZ4_DECODE;
					q = qstop;
				}
				if (q >= qstop) {
					asm volatile("" ::: "memory");
					uint8_t *mp = W + mdst;
					const uint8_t *fp = W + s0;
#ifndef FAST_EXP_NOCOPY	/* (what-if, wrong output: the flag protocol alone) */
					if (off >= mlen) {
						/* Source and destination do not overlap.  ONE LDS round trip per 64
						 * bytes: every load is issued before anything is stored, and the
						 * ragged end is one more 8-byte (or 4-byte) copy placed so that it
						 * ENDS with the match -- overlapping stores instead of a cascade of
						 * 4/2/1-byte ones.  Loads may run past the source into later window
						 * bytes or the slack behind the window; stores never pass the match. */
						/* An unaligned LDS access costs one cycle per ACTIVE LANE whatever its width
						 * (tools/ubench_lds.hip), and this phase runs at the speed of the LDS pipe: so
						 * every lane moves its match with as few, as wide accesses as it takes --
						 * 16-byte pieces from 16 bytes up (the last one placed so that it ENDS with the
						 * match: overlapping stores instead of a ragged tail), two 8-byte ones below that,
						 * one 8-byte load and two 4-byte stores below 8.  All loads of a step are issued
						 * before anything is stored.  Nothing is read outside the source from 8 bytes up. */
						if (mlen >= 16) {
							for (uint32_t i = 0;; i += 32) {	/* one trip up to 47 bytes */
								const uint32_t left = mlen - i;
								const uint4 v0 = lds_ld16(fp + i);
								uint4 v1 = make_uint4(0, 0, 0, 0), vt = make_uint4(0, 0, 0, 0);	/* (not from v0: nothing makes one DS read wait for another) */
								if (left >= 32)
									v1 = lds_ld16(fp + i + 16);
								const bool last = left < 48;	/* 16..47 bytes left: this trip ends the match */
								if (last && (left & 15))
									vt = lds_ld16(fp + mlen - 16);
								lds_st16(mp + i, v0);
								if (left >= 32)
									lds_st16(mp + i + 16, v1);
								if (last) {
									if (left & 15)
										lds_st16(mp + mlen - 16, vt);
									break;
								}
							}
						} else if (mlen >= 8) {
							const uint64_t a0 = lds_ld8(fp), at = lds_ld8(fp + mlen - 8);
							lds_st8(mp, a0);
							if (mlen != 8)
								lds_st8(mp + mlen - 8, at);
						} else {
							const uint64_t a0 = lds_ld8(fp);	/* (may read up to 7 bytes past the source: window bytes or slack) */
							if (mlen < 4) {
								/* 1..3 bytes: only the deflate front end makes these (LZ4 matches are
								 * at least four bytes long) */
								lds_st_tail(mp, a0, mlen);
							} else {
								/* 4 <= mlen <= 7: two overlapping 4-byte stores */
								lds_st4(mp, (uint32_t)a0);
								if (mlen != 4)
									lds_st4(mp + mlen - 4, (uint32_t)(a0 >> (8 * (mlen - 4))));
							}
						}
					} else if (off >= 8) {
						/* overlapping, period >= 8: a forward 8-byte copy only reads bytes
						 * that this thread stored at least one iteration earlier */
						uint32_t i = 0;
						for (; i + 8 <= mlen; i += 8) {
							uint64_t a = lds_ld8(mp + i - off);
							lds_st8(mp + i, a);
							asm volatile("" ::: "memory");	/* keep load/store order: the ranges overlap */
						}
						for (; i < mlen; i++)
							{ uint8_t bq = mp[(int)i - (int)off]; asm volatile("" ::: "memory"); mp[i] = bq; asm volatile("" ::: "memory"); }
					} else {
						/* overlapping match: replicate with period `off`; every byte
						 * read is one this thread (or an earlier sequence) already wrote */
						for (uint32_t i = 0; i < mlen; i++)
							{ uint8_t bq = mp[(int)i - (int)off]; asm volatile("" ::: "memory"); mp[i] = bq; asm volatile("" ::: "memory"); }
					}
#else
					(void)mp; (void)fp;
#endif
					asm volatile("" ::: "memory");
					atomicOr(&donebits[k >> 5], 1u << (k & 31));
					fin = true;
				}
			}
#if defined(LA_DIAG) && !defined(LA_DIAG_L) && !defined(LA_DIAG_M)
			if (lane == 0 && la_diag_stamps) atomicAdd(&la_diag_stamps[(size_t)blockIdx.x * 8 + 7], __builtin_readcyclecounter() - t_it0);
#endif
			if (__ballot(!fin) == 0)
				break;
			__builtin_amdgcn_s_sleep(POLL_SLEEP);	/* back off: polling waves share the LDS with copying ones */
		}
	}
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

/* blocks the SEG launch must take: eligible, with a table, more than MAXSEQ sequences */
__global__ __launch_bounds__(256) void lz4_classify_kernel(const la_lz4_block *__restrict__ blocks, uint32_t n,
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

#ifdef LA_DIAG
extern "C" int la_diag_set_stamps(void *d_buf)
{
	unsigned long long *p = (unsigned long long *)d_buf;
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(la_diag_stamps), &p, sizeof(p));
}
#endif

void la_launch_lz4_expand_fast(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t long_thr)
{
	if (n == 0) return;
	hipLaunchKernelGGL((lz4_expand_fast_kernel<LA_LZ4_FAST_MAXSEQ, false>), dim3(n), dim3(FAST_THREADS), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq,
	    d_table, d_table_off, (const uint32_t *)nullptr, (const uint32_t *)nullptr, long_thr);
}

/* blocks with more than LA_LZ4_FAST_MAXSEQ sequences, over the WHOLE table: classify + a small
 * grid that shares them out.  d_big: n + 1 words of workspace (count first). */
void la_launch_lz4_expand_fast_big(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t *d_big, uint32_t long_thr)
{
	if (n == 0) return;
	(void)hipMemsetAsync(d_big, 0, sizeof(uint32_t), s);
	hipLaunchKernelGGL(lz4_classify_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_blocks, n, d_status, d_nseq,
	    d_big + 1, d_big, d_out_len, long_thr);
	const uint32_t grid = n < 1024u ? n : 1024u;
	hipLaunchKernelGGL((lz4_expand_fast_kernel<LA_LZ4_FAST_MAXSEQ, true>), dim3(grid), dim3(FAST_THREADS), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq,
	    d_table, d_table_off, (const uint32_t *)(d_big + 1), (const uint32_t *)d_big, long_thr);
}
