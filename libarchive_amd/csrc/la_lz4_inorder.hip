/*
 * la_lz4_inorder.hip -- in-order LDS-window LZ4 expand kernel (gfx950): the hot kernel of the
 * lz4 filter path for independent blocks of at most 64 KiB
 * (libarchive/archive_read_support_filter_lz4.c:557-561, the LZ4_decompress_safe call of BD=4
 * frames), and the second phase of the two-phase inflate (archive_read_support_filter_gzip.c:479).
 *
 * Round 3.  The round-1/2 kernel (la_lz4_fast.hip, kept as the cross-check) gave every sequence to a
 * thread and let matches POLL per-sequence done bits: 31 poll iterations per wave and block, each one
 * running the whole (divergent) copy code for a handful of ready lanes -- 47 k instructions per block, half
 * of the wave cycles waiting.  This kernel has no flags per sequence, no search and no polling in the
 * match phase at all.  It rests on one hardware fact: the LDS serves the DS instructions of ONE wave in
 * program order.  So a single wave that takes the block's sequences IN STREAM ORDER, 64 at a time (one
 * per lane), sees every byte that an earlier group produced without any synchronisation; only a match
 * whose source reaches into its OWN group has to wait, and those (about one lane in twenty on the C2
 * stream) are finished in a few more passes over the same registers.
 *
 * One workgroup = three waves with fixed roles that share a 64 KiB LDS window (two workgroups per CU),
 * persistent over IO_BPW consecutive blocks:
 *
 *   wave 1, L (literals)   runs AHEAD of the matcher: per group of 64 sequences it loads the table
 *       entries (coalesced 8 bytes per lane, two groups in flight) and each lane's literal run straight
 *       from the compressed payload (two unaligned 16-byte global loads, requested one group ahead),
 *       stores it into the window with exact-length LDS stores, and bumps `lit_total`.
 *   wave 0, M (matches)    per group: waits (normally not at all) until L has passed the group, then
 *       pass 1 copies every match whose source ends before the group's first match -- all loads of a lane
 *       before its stores, 16 bytes per LDS access -- and the few remaining lanes follow in rounds: a
 *       lane is ready once its source ends before the match of the lowest unfinished lane (everything
 *       below that is final).  Overlapping matches (offset < length) double their period in place.
 *       After each group M publishes the window position up to which everything is final.
 *   wave 2, F (flush)      trails M: copies the final part of the window to the decoded slab in
 *       1 KiB steps (16-byte aligned LDS reads, coalesced 16-byte stores; window and slab addresses are
 *       congruent modulo 16) and hands the window back to L when the block is out.
 *
 * While M works through block j, L has already fetched the descriptor, the first table groups and the
 * first literal bytes of block j+1 and waits only for F's hand-back: HBM latency is off the critical
 * path, which is M's instruction stream.  No __syncthreads() after the start, every spin is bounded and
 * watches a common abort word, so a table that does not add up fails the block instead of hanging.
 *
 * HBM traffic per block: payload once (only literal bytes are touched, every cache line once or twice
 * from L1/L2), the sequence table twice (L and M, the second time from L2), decoded bytes once out.
 */
#include "la_dev.h"

#ifndef IO_BPW
#define IO_BPW 8u		/* consecutive blocks one workgroup takes */
#endif
#define IO_THREADS 192
#define IO_SPIN_LIMIT (1u << 22)
#ifndef IO_MAX_EXACT
#define IO_MAX_EXACT 8		/* unfinished lanes after pass 1 up to which the exact dependency masks are built */
#endif

struct io_ctl {
	uint32_t lit_total;	/* groups of 64 sequences whose literals are in the window (monotonic over the workgroup's blocks) */
	uint32_t match_flag;	/* (processed-block counter & 0x7FFF) << 17 | window position up to which all bytes are final */
	uint32_t flushed;	/* processed blocks whose window has been read out completely */
	uint32_t abort;		/* a wave gave up: everybody leaves */
};

#ifdef LA_DIAG
__device__ unsigned long long *la_diag_io_stamps;
#define IO_STAMP_ADD(slot, v)                                                                     \
	do {                                                                                      \
		if (lane == 0 && la_diag_io_stamps)                                               \
			la_diag_io_stamps[(size_t)blockIdx.x * 16 + (slot)] += (unsigned long long)(v); \
	} while (0)
#define IO_NOW() __builtin_readcyclecounter()
#else
#define IO_STAMP_ADD(slot, v) do { } while (0)
#define IO_NOW() 0ull
#endif

typedef uint64_t seq_t;
#define SEQ_LIT_SRC(e) ((uint32_t)((e) & 0xFFFFu))
#define SEQ_LIT_LEN(e) ((uint32_t)(((e) >> 16) & 0xFFFFu))
#define SEQ_DST(e)     ((uint32_t)(((e) >> 32) & 0xFFFFu))
#define SEQ_OFF(e)     ((uint32_t)((e) >> 48))

__device__ __forceinline__ seq_t io_ent(const la_lz4_seq *t, uint32_t k, uint32_t ns)
{
	return k < ns ? *(const uint64_t *)(const void *)(t + k) : 0ull;
}

__device__ __forceinline__ uint32_t io_ld(const uint32_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void io_st(uint32_t *p, uint32_t v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* wait until *p >= want (wrap-safe); false when the workgroup aborts or the spin limit is hit */
__device__ __forceinline__ bool io_wait_ge(const uint32_t *p, uint32_t want, io_ctl *ctl, uint32_t sleep)
{
	uint32_t spins = 0;
	while ((int32_t)(io_ld(p) - want) < 0) {
		if (++spins > IO_SPIN_LIMIT || io_ld(&ctl->abort))
			return false;
		if (sleep)
			__builtin_amdgcn_s_sleep(2);
	}
	return true;
}

/* what every wave needs to know about a block; all three waves evaluate the same predicate on the
 * same words, so they agree on which blocks are taken */
struct io_blk {
	bool take;
	uint32_t ns, olen;
	const la_lz4_seq *tab;
	const uint8_t *s;	/* payload */
	uint64_t s_room;	/* bytes of the image from s on */
	uint8_t *g_out;
};

__device__ __forceinline__ io_blk io_load_blk(uint32_t bi, uint32_t n, const uint8_t *src, uint64_t src_bytes,
    const la_lz4_block *blocks, uint8_t *dst, uint64_t dst_cap, const uint64_t *dst_off, const uint32_t *out_len,
    const uint32_t *status, const uint32_t *nseq, const la_lz4_seq *table, const uint64_t *table_off, uint32_t long_thr)
{
	io_blk r;
	r.take = false;
	r.ns = r.olen = 0;
	r.tab = nullptr; r.s = nullptr; r.s_room = 0; r.g_out = nullptr;
	if (bi >= n)
		return r;
	const la_lz4_block b = blocks[bi];
	const uint32_t olen = out_len[bi], ns = nseq[bi];
	const uint64_t doff = dst_off[bi];
	/* same predicate as the general kernel's skip test (0xFFFFFFFF: the block has no table) */
	if (status[bi] != LA_ST_OK || olen == 0 || olen > 65536u || !la_lz4_fast_eligible(b) || ns == 0xFFFFFFFFu || ns == 0 ||
	    doff + olen > dst_cap || b.src_off > src_bytes || la_lz4_long_sequences(ns, olen, long_thr))
		return r;
	r.take = true;
	r.ns = ns;
	r.olen = olen;
	r.tab = table + table_off[bi];
	r.s = src + b.src_off;
	r.s_room = src_bytes - b.src_off;
	r.g_out = dst + doff;
	return r;
}

/* exact-length copy inside the window, ranges do not overlap (n <= distance), n >= 1.  Every load of a
 * trip is issued before its stores; 16-byte pieces from 16 bytes up with the last one placed so that it
 * ENDS with the copy (overlapping stores instead of a ragged tail), two 8-byte accesses below that, two
 * 4-byte ones below 8: an unaligned LDS access costs one cycle per active lane whatever its width
 * (profiles/r02_ubench_lds.txt).  Loads may run up to 15 bytes past the source (window bytes or the slack
 * behind the window); stores never pass d + n. */
__device__ __forceinline__ void io_copy_exact(uint8_t *d, const uint8_t *s, uint32_t n)
{
	if (n >= 16) {
		for (uint32_t i = 0;; i += 32) {	/* one trip up to 47 bytes */
			const uint32_t left = n - i;
			const uint4 v0 = lds_ld16(s + i);
			uint4 v1 = v0, vt = v0;
			if (left >= 32)
				v1 = lds_ld16(s + i + 16);
			const bool last = left < 48;
			if (last && (left & 15))
				vt = lds_ld16(s + n - 16);
			lds_st16(d + i, v0);
			if (left >= 32)
				lds_st16(d + i + 16, v1);
			if (last) {
				if (left & 15)
					lds_st16(d + n - 16, vt);
				break;
			}
		}
	} else if (n >= 8) {
		const uint64_t a0 = lds_ld8(s), at = lds_ld8(s + n - 8);
		lds_st8(d, a0);
		if (n != 8)
			lds_st8(d + n - 8, at);
	} else {
		const uint64_t a0 = lds_ld8(s);
		if (n < 4) {
			lds_st_tail(d, a0, n);	/* 1..3 bytes: only the deflate front end makes these */
		} else {
			lds_st4(d, (uint32_t)a0);
			if (n != 4)
				lds_st4(d + n - 4, (uint32_t)(a0 >> (8 * (n - 4))));
		}
	}
}

/* one match: mlen bytes to mp from fp = mp - off.  A match that overlaps its source (off < mlen) is a
 * periodic run: the valid stretch behind fp doubles with every step (off, 2 off, 4 off, ... bytes are
 * copied from fp itself), each step a non-overlapping copy that the in-order LDS lets follow the previous
 * one without a wait.  The usual match is one step. */
__device__ __forceinline__ void io_copy_match(uint8_t *mp, uint32_t off, uint32_t mlen)
{
	const uint8_t *fp = mp - off;
	uint32_t done = 0;
	do {
		uint32_t nn = off + done;	/* bytes valid from fp on */
		if (nn > mlen - done)
			nn = mlen - done;
		io_copy_exact(mp + done, fp, nn);
		asm volatile("" ::: "memory");	/* keep the order: the next step reads what this one stored */
		done += nn;
	} while (done < mlen);
}

/* exact-length literal store of 1..32 bytes: p0 = payload bytes [0, 16), pt = bytes [ll - 16, ll) when
 * ll > 16 */
__device__ __forceinline__ void io_lit_store(uint8_t *d, const uint4 p0, const uint4 pt, uint32_t ll)
{
	const uint64_t lo = ((uint64_t)p0.y << 32) | p0.x, hi = ((uint64_t)p0.w << 32) | p0.z;
	if (ll >= 16) {
		lds_st16(d, p0);
		if (ll > 16)
			lds_st16(d + ll - 16, pt);
	} else if (ll >= 8) {
		lds_st8(d, lo);
		if (ll != 8) {
			const uint32_t t = 8 * (ll - 8);	/* 8 .. 56 */
			lds_st8(d + ll - 8, (lo >> t) | (hi << (64 - t)));
		}
	} else if (ll >= 4) {
		lds_st4(d, p0.x);
		if (ll != 4)
			lds_st4(d + ll - 4, (uint32_t)(lo >> (8 * (ll - 4))));
	} else {
		lds_st_tail(d, lo, ll);
	}
}

/* 16 payload bytes at p; the last few bytes of the image are assembled byte by byte (never read past it) */
__device__ __forceinline__ uint4 io_ld_payload16(const uint8_t *s, uint32_t at, uint64_t s_room)
{
	if ((uint64_t)at + 16 <= s_room)
		return ld_u128(s + at);
	uint32_t w[4] = { 0, 0, 0, 0 };
	for (uint32_t i = 0; i < 16 && (uint64_t)at + i < s_room; i++)
		w[i >> 2] |= (uint32_t)s[at + i] << (8 * (i & 3));
	return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(IO_THREADS) void lz4_expand_inorder_kernel(
    const uint8_t *__restrict__ src, uint64_t src_bytes, const la_lz4_block *__restrict__ blocks,
    uint32_t n, uint8_t *__restrict__ dst, uint64_t dst_cap, const uint64_t *__restrict__ dst_off,
    const uint32_t *__restrict__ out_len, uint32_t *status_out,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table,
    const uint64_t *__restrict__ table_off, uint32_t long_thr)
{
	const uint32_t *status = status_out;
	/* 16 bytes of headroom + up to 15 of alignment shift + the window + slack for over-reads */
	__shared__ __attribute__((aligned(16))) uint8_t win[16 + 16 + 65536 + 96];
	__shared__ io_ctl ctl_s;
	io_ctl *const ctl = &ctl_s;

	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0) {
		ctl->lit_total = 0;
		ctl->match_flag = 0x7FFFu << 17;	/* no block has this number first */
		ctl->flushed = 0;
		ctl->abort = 0;
	}
	__syncthreads();	/* the only barrier: the roles part here */

	const uint32_t b_first = blockIdx.x * IO_BPW;

#define IO_LOAD_BLK(bi_) io_load_blk((bi_), n, src, src_bytes, blocks, dst, dst_cap, dst_off, out_len, status, nseq, table, table_off, long_thr)

	if (wave == 0) {
		/* ================= M: matches, in stream order ================= */
		__builtin_amdgcn_s_setprio(3);
		uint32_t gbase = 0, seqno = 0;
		for (uint32_t it = 0; it < IO_BPW; it++) {
			const uint32_t bi = b_first + it;
			const io_blk B = IO_LOAD_BLK(bi);
			if (!B.take)
				continue;
			uint8_t *const W = win + 16 + ((uintptr_t)B.g_out & 15);	/* W[i] <-> g_out[i], congruent mod 16 */
			const uint32_t ns = B.ns, olen = B.olen;
			const uint32_t ng = (ns + 63) >> 6;
			seq_t e0 = io_ent(B.tab, lane, ns), e1 = io_ent(B.tab, 64 + lane, ns);
			const unsigned long long t_blk0 = IO_NOW();
			for (uint32_t g = 0; g < ng; g++) {
				const seq_t e2 = io_ent(B.tab, (g + 2) * 64 + lane, ns);
				const uint32_t k = g * 64 + lane;
				const bool valid = k < ns;
				/* where the next group's output begins = where this group's ends */
				const uint32_t s_next = (g + 1) * 64 < ns ? (uint32_t)__builtin_amdgcn_readfirstlane((int)SEQ_DST(e1)) : olen;
				const uint32_t d = valid ? SEQ_DST(e0) : olen;
				const uint32_t ll = SEQ_LIT_LEN(e0), off = SEQ_OFF(e0);
				const uint32_t mdst = d + ll;
				/* output position of the next sequence: lane + 1's, the next group's first for lane 63 */
				uint32_t nd = (uint32_t)__builtin_amdgcn_update_dpp((int)s_next, (int)d, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
				if (k + 1 >= ns)
					nd = olen;	/* the last sequence ends with the block */
				uint32_t mlen = (valid && nd > mdst && nd <= olen) ? nd - mdst : 0;
				/* a table entry that does not add up (impossible for one the parse kernel wrote) */
				const bool bad = valid && (mdst > olen || nd < mdst || (mlen != 0 && (off == 0 || off > mdst)));
				if (__ballot(bad) != 0) {
					if (lane == 0) {
						status_out[bi] = LA_ST_LZ4_DECODE;
						io_st(&ctl->abort, 1);
					}
					return;
				}
				const uint32_t s0 = mdst - off;
				const uint32_t send = s0 + (mlen < off ? mlen : off);	/* first byte behind the part of the source that exists before the copy starts */
				bool fin = mlen == 0;

				/* L has to be past this group (it normally is, by a group or more) */
				const unsigned long long t_w0 = IO_NOW();
				if (!io_wait_ge(&ctl->lit_total, gbase + g + 1, ctl, 0)) {
					if (lane == 0) {
						status_out[bi] = LA_ST_LZ4_DECODE;
						io_st(&ctl->abort, 1);
					}
					return;
				}
				IO_STAMP_ADD(1, IO_NOW() - t_w0);
				asm volatile("" ::: "memory");

				/* pass 1: every match whose source lies in front of the group's first match.  Rounds: a
				 * lane is ready when its source ends in front of the match of the lowest unfinished lane */
				uint32_t P = (uint32_t)__builtin_amdgcn_readfirstlane((int)mdst);
				uint32_t rounds = 0;
				for (;;) {
					const bool ready = !fin && send <= P;
					if (ready)
						io_copy_match(W + mdst, off, mlen);
					fin = fin || ready;
					asm volatile("" ::: "memory");
					const uint64_t unf = __ballot(!fin);
					if (unf == 0)
						break;
					const uint32_t low = (uint32_t)__builtin_ctzll(unf);
					P = (uint32_t)__builtin_amdgcn_readlane((int)mdst, (int)low);
					if (++rounds > 64u) {	/* cannot happen: the lowest unfinished lane is always ready */
						if (lane == 0) {
							status_out[bi] = LA_ST_LZ4_DECODE;
							io_st(&ctl->abort, 1);
						}
						return;
					}
				}
				IO_STAMP_ADD(2, rounds + 1);
				/* everything below s_next is final */
				asm volatile("" ::: "memory");
				if (lane == 0)
					io_st(&ctl->match_flag, ((seqno & 0x7FFFu) << 17) | s_next);
				e0 = e1;
				e1 = e2;
			}
			IO_STAMP_ADD(0, IO_NOW() - t_blk0);
			IO_STAMP_ADD(3, ng);
			gbase += ng;
			seqno++;
		}
	} else if (wave == 1) {
		/* ================= L: literals, one group (or more) ahead of M ================= */
		uint32_t gbase = 0, seqno = 0;
		for (uint32_t it = 0; it < IO_BPW; it++) {
			const uint32_t bi = b_first + it;
			const io_blk B = IO_LOAD_BLK(bi);
			if (!B.take)
				continue;
			uint8_t *const W = win + 16 + ((uintptr_t)B.g_out & 15);
			const uint32_t ns = B.ns;
			const uint32_t ng = (ns + 63) >> 6;
			const uint8_t *const s = B.s;
			const uint64_t s_room = B.s_room;
			seq_t e0 = io_ent(B.tab, lane, ns), e1 = io_ent(B.tab, 64 + lane, ns);
			/* the first group's literal bytes are requested before the window is even free */
			uint4 p0 = make_uint4(0, 0, 0, 0), pt = p0;
			{
				const uint32_t ls = SEQ_LIT_SRC(e0), ll = SEQ_LIT_LEN(e0);
				if (ll) {
					p0 = io_ld_payload16(s, ls, s_room);
					if (ll > 16)
						pt = io_ld_payload16(s, ls + (ll > 32 ? 32u : ll) - 16, s_room);
				}
			}
			/* the window is ours when F has read the previous block out */
			const unsigned long long t_w0 = IO_NOW();
			if (!io_wait_ge(&ctl->flushed, seqno, ctl, 1)) {
				if (lane == 0) {
					status_out[bi] = LA_ST_LZ4_DECODE;
					io_st(&ctl->abort, 1);
				}
				return;
			}
			IO_STAMP_ADD(5, IO_NOW() - t_w0);
			asm volatile("" ::: "memory");
			for (uint32_t g = 0; g < ng; g++) {
				const seq_t e2 = io_ent(B.tab, (g + 2) * 64 + lane, ns);
				/* next group's literal bytes: in flight while this group is stored */
				uint4 q0 = make_uint4(0, 0, 0, 0), qt = q0;
				{
					const uint32_t ls1 = SEQ_LIT_SRC(e1), ll1 = SEQ_LIT_LEN(e1);
					if (ll1) {
						q0 = io_ld_payload16(s, ls1, s_room);
						if (ll1 > 16)
							qt = io_ld_payload16(s, ls1 + (ll1 > 32 ? 32u : ll1) - 16, s_room);
					}
				}
				const uint32_t ls = SEQ_LIT_SRC(e0), ll = SEQ_LIT_LEN(e0), d = SEQ_DST(e0);
				/* (entries beyond the table are 0: ll = 0.)  A run that would leave the window or the
				 * payload is not stored: M fails such a block by its own checks or the hash does */
				const bool okl = ll != 0 && d + ll <= 65536u && (uint64_t)ls + ll <= s_room;
				if (okl) {
					uint8_t *wd = W + d;
					if (ll <= 32) {
						io_lit_store(wd, p0, pt, ll);
					} else {
						/* long run: [0, 16) and [16, 32) are here already, the rest follows 16 bytes at a
						 * time, the last piece placed so that it ends with the run */
						lds_st16(wd, p0);
						lds_st16(wd + 16, pt);
						uint32_t i = 32;
						for (; i + 16 <= ll; i += 16)
							lds_st16(wd + i, ld_u128(s + ls + i));
						if (i < ll)
							lds_st16(wd + ll - 16, ld_u128(s + ls + ll - 16));
					}
				}
				asm volatile("" ::: "memory");
				if (lane == 0)
					io_st(&ctl->lit_total, gbase + g + 1);
				e0 = e1;
				e1 = e2;
				p0 = q0;
				pt = qt;
			}
			IO_STAMP_ADD(4, ng);
			gbase += ng;
			seqno++;
		}
	} else {
		/* ================= F: flush, behind M ================= */
		uint32_t seqno = 0;
		for (uint32_t it = 0; it < IO_BPW; it++) {
			const uint32_t bi = b_first + it;
			const io_blk B = IO_LOAD_BLK(bi);
			if (!B.take)
				continue;
			uint8_t *const g_out = B.g_out;
			const uint8_t *const W = win + 16 + ((uintptr_t)g_out & 15);
			const uint32_t olen = B.olen;
			uint32_t head = (16u - (uint32_t)((uintptr_t)g_out & 15)) & 15u;
			if (head > olen)
				head = olen;
			const uint32_t nunits = (olen - head) >> 4;	/* aligned 16-byte units behind the head */
			const uint4 *const wsrc = (const uint4 *)(W + head);
			uint4 *const gdst = (uint4 *)(g_out + head);
			uint32_t done = 0, spins = 0;
			for (;;) {
				const uint32_t flag = io_ld(&ctl->match_flag);
				const uint32_t pos = (flag >> 17) == (seqno & 0x7FFFu) ? (flag & 0x1FFFFu) : 0u;
				asm volatile("" ::: "memory");
				const bool complete = pos >= olen;
				uint32_t avail = pos > head ? (pos - head) >> 4 : 0u;
				if (avail > nunits)
					avail = nunits;
				bool moved = false;
				while (avail - done >= 64u || (complete && done < avail)) {
					const uint32_t u = done + lane;
					if (u < avail)
						gdst[u] = wsrc[u];
					done += 64u;
					if (done > avail)
						done = avail;
					moved = true;
				}
				if (complete) {
					if (lane < head)
						g_out[lane] = W[lane];
					const uint32_t tail0 = head + (nunits << 4);
					if (tail0 + lane < olen)
						g_out[tail0 + lane] = W[tail0 + lane];
					break;
				}
				if (moved) {
					spins = 0;
				} else if (++spins > IO_SPIN_LIMIT || io_ld(&ctl->abort)) {
					if (lane == 0) {
						status_out[bi] = LA_ST_LZ4_DECODE;
						io_st(&ctl->abort, 1);
					}
					return;
				}
				__builtin_amdgcn_s_sleep(4);
			}
			/* every LDS read of this block has returned (its data went into the stores above) */
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			seqno++;
			if (lane == 0)
				io_st(&ctl->flushed, seqno);
		}
	}
#undef IO_LOAD_BLK
}

#ifdef LA_DIAG
extern "C" int la_diag_set_io_stamps(void *d_buf)
{
	unsigned long long *p = (unsigned long long *)d_buf;
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(la_diag_io_stamps), &p, sizeof(p));
}
#endif

void la_launch_lz4_expand_inorder(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t long_thr)
{
	if (n == 0) return;
	const uint32_t grid = (n + IO_BPW - 1) / IO_BPW;
	hipLaunchKernelGGL(lz4_expand_inorder_kernel, dim3(grid), dim3(IO_THREADS), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq,
	    d_table, d_table_off, long_thr);
}
