/*
 * la_lz4_inorder.hip -- in-order LDS-window LZ4 expand kernel (gfx950), round 3: a SECOND implementation of the
 * expand step of the lz4 filter path for independent blocks of at most 64 KiB
 * (libarchive/archive_read_support_filter_lz4.c:557-561, the LZ4_decompress_safe call of BD=4 frames) and of the
 * second phase of the two-phase inflate (archive_read_support_filter_gzip.c:479).  Selected with
 * LA_LZ4_OPT_EXPAND_INORDER / LA_GZ_OPT_EXPAND_INORDER; every lz4 / deflate GPU test runs it beside the default
 * kernel (la_lz4_fast.hip) and requires identical bytes.  It is NOT the default: on the C2 stream it reaches
 * 0.10 - 0.12 of the 8 TB/s roofline against 0.14 for the polling kernel (profiles/r03_inorder_experiments.md has
 * every measurement quoted below).
 *
 * Idea.  The polling kernel gives every sequence to a thread and lets matches POLL per-sequence done bits: 31 poll
 * iterations per wave and block, each running the divergent copy code for a handful of ready lanes.  This kernel has
 * no flags per sequence and no polling in the match phase.  It rests on one hardware fact: the LDS executes the DS
 * instructions of ONE wave in program order.  A single wave that takes the block's sequences IN STREAM ORDER, 64 at
 * a time (one per lane), sees every byte an earlier group produced without any synchronisation; only a match whose
 * source reaches into a match of its OWN group has to wait, and those (about one lane in eighteen on C2) follow in
 * one or two more passes over the same registers.
 *
 * One workgroup = 64 KiB LDS window + fixed roles (two workgroups per CU), persistent over IO_BPW consecutive blocks:
 *   waves 1..IO_LWAVES, L (literals + planning)  run AHEAD of the matcher; the groups of 64 sequences are shared
 *       out by their running number.  Per group: table entries (coalesced 8 bytes per lane) two groups ahead,
 *       each lane's literal run straight from the compressed payload (two unaligned 16-byte global loads, one group
 *       ahead, addresses clamped into the image instead of predicated so that no load is waited for behind its
 *       issue), exact-length LDS stores of the literals, then for every match: where, how long, from where, and
 *       WHICH lanes of the same group its source overlaps (two binary searches over the lanes, ds_bpermute) --
 *       written as one tagged 16-byte record per sequence into a ring (the tag is the group's running number: all
 *       64 records appear with one ds_write_b128, after the group's literals).
 *   wave 0, M (matches)  per group: the records (requested with the previous group's loads; the tag says whether
 *       they are there), eleven length-class masks, then pass 1 = every match without an in-group dependency as ONE
 *       straight line of predicated DS instructions (io_copy_fast, la_dev.h), then rounds: a lane is ready when none
 *       of the lanes it waits for is unfinished.  Matches of more than 64 bytes or overlapping their source take the
 *       generic per-lane path (period doubling).  After each group M publishes the window position up to which
 *       everything is final, and the group count (the ring slot is free).
 *   last wave, F (flush)  trails M: copies the final part of the window to the decoded slab in 4 KiB steps
 *       (16-byte aligned LDS reads, coalesced 16-byte stores; window and slab addresses are congruent modulo 16)
 *       and hands the window back to L when the block is out.
 * No __syncthreads() after the start; every spin is bounded and watches a common abort word; a wave that gives up
 * marks the block and lets the workgroup run out (no early return: see IO_FAIL).
 *
 * What was measured (MI355X, C2 blocks: 32.3 groups of 64 sequences per block, 2.2 passes per group):
 *   - M needs about 2 800 - 3 000 cycles per group: records + masks + publish about 400, pass 1 about 600 - 850,
 *     the 1.2 later rounds about 1 000 each.  Two blocks per CU: 0.10 - 0.12 of the roofline.
 *   - The matcher is ONE wave, and a lone wave is bound by the latency of its own instruction stream: about eight
 *     cycles per dependent instruction, a DS instruction every 15 - 30 cycles WHATEVER the number of lanes it enables
 *     (profiles/r02_ubench_lds.txt "1 wave"), an LDS round trip of 63 cycles.  A pass of sixteen predicated DS
 *     instructions therefore costs the same for three lanes as for sixty.  Pass 1 (dense) is efficient, the later
 *     rounds (three lanes each) are not, and they are on the critical path of every group.
 *   - What did not help: skipping empty classes with branches (a taken branch costs what an empty DS instruction
 *     costs), 2 / 6 / 14 literal waves (M's time per group 2 900 / 3 000 / 3 700: more resident waves slow every wave),
 *     no priority for M, the literal stores or the flush switched off (no change: M is the bottleneck), copying the
 *     dependent matches one by one with the whole wave instead of rounds (400 cycles per match, 5.25 per group).
 *   - What would: the later rounds on a second matcher wave beside pass 1 of the next group (needs the dependency
 *     of every match on the PREVIOUS group's late matches from L as well), or several matcher waves per block with a
 *     completed-prefix counter; both estimated at 1 400 - 1 800 cycles per group (0.2 - 0.25), not built.
 * hipcc 7.2 notes: a workgroup of six waves made the compiler pad the register allocation to 129 ("3 waves per SIMD"
 * derived from the LDS size) and only ONE workgroup fitted a CU -- the workgroup is launched with eight waves, two
 * leave at once; an early `return` inside the literal waves' group loop was miscompiled (loop-carried prefetch
 * registers restored from the wrong set after the ring wait; found by the reference's test_compat_lz4_B4 fixture).
 */
#include "la_dev.h"

#ifndef IO_BPW
#define IO_BPW 8u		/* consecutive blocks one workgroup takes */
#endif
#ifndef IO_LWAVES
#define IO_LWAVES 6u		/* literal waves per workgroup */
#endif
/* M + L waves + F work; the workgroup is launched with eight waves all the same (the rest leave at once): with two
 * workgroups of SIX waves per CU the compiler derives "3 waves per SIMD" from the LDS size and pads the register
 * allocation to 129 so that no fourth wave fits -- and then the second workgroup's 2+2+1+1 waves do not fit beside
 * the first's (measured: one workgroup per CU).  Eight waves per workgroup make it 4 per SIMD, 2 per SIMD and group. */
#define IO_THREADS 512u
#define IO_SPIN_LIMIT (1u << 22)
#ifndef IO_RING
#define IO_RING 8u		/* groups of 64 sequence records L may be ahead of M */
#endif

struct io_ctl {
	uint32_t match_groups;	/* groups M is done with (same count) */
	uint32_t match_flag;	/* (processed-block counter & 0x7FFF) << 17 | window position up to which all bytes are final */
	uint32_t flushed;	/* processed blocks whose window has been read out completely */
	uint32_t abort;		/* a wave gave up: everybody leaves */
};

#ifdef LA_DIAG
__device__ unsigned long long *la_diag_io_stamps;
/* counters are kept in registers and written out once per workgroup (a read-modify-write of global
 * memory per stamp would cost more than what it measures) */
#define IO_STAMP_DECL unsigned long long io_acc[16] = { 0 }
#define IO_STAMP_ADD(slot, v) (io_acc[slot] += (unsigned long long)(v))
#define IO_STAMP_FLUSH()                                                                          \
	do {                                                                                      \
		if (lane == 0 && la_diag_io_stamps)                                               \
			for (int q_ = 0; q_ < 16; q_++)                                           \
				if (io_acc[q_])                                                   \
					atomicAdd(&la_diag_io_stamps[(size_t)blockIdx.x * 16 + q_], io_acc[q_]); \
	} while (0)
#define IO_NOW() __builtin_readcyclecounter()
#else
#define IO_STAMP_DECL do { } while (0)
#define IO_STAMP_ADD(slot, v) do { } while (0)
#define IO_STAMP_FLUSH() do { } while (0)
#define IO_NOW() 0ull
#endif

typedef uint64_t seq_t;
#define SEQ_LIT_SRC(e) ((uint32_t)((e) & 0xFFFFu))
#define SEQ_LIT_LEN(e) ((uint32_t)(((e) >> 16) & 0xFFFFu))
#define SEQ_DST(e)     ((uint32_t)(((e) >> 32) & 0xFFFFu))
#define SEQ_OFF(e)     ((uint32_t)((e) >> 48))

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
	/* (the words are the same in every lane: readfirstlane tells the compiler so, and the loop is scalar) */
	while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)io_ld(p)) - want) < 0) {
		if (++spins > IO_SPIN_LIMIT || __builtin_amdgcn_readfirstlane((int)io_ld(&ctl->abort)))
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
 * trip is issued before its stores (the conditional loads start from zeros, not from the first load's
 * value: nothing makes one DS read wait for another); 16-byte pieces from 16 bytes up with the last one
 * placed so that it ENDS with the copy (overlapping stores instead of a ragged tail), two 8-byte accesses
 * below that, two 4-byte ones below 8: an unaligned LDS access costs one cycle per active lane whatever
 * its width (profiles/r02_ubench_lds.txt).  Loads may run up to 15 bytes past the source (window bytes or
 * the slack behind the window); stores never pass d + n. */
__device__ __forceinline__ void io_copy_exact(uint8_t *d, const uint8_t *s, uint32_t n)
{
	if (n >= 16) {
		for (uint32_t i = 0;; i += 32) {	/* one trip up to 47 bytes */
			const uint32_t left = n - i;
			const bool two = left >= 32, last = left < 48, rag = last && (left & 15);
			const uint4 v0 = lds_ld16(s + i);
			uint4 v1 = make_uint4(0, 0, 0, 0), vt = make_uint4(0, 0, 0, 0);
			if (two)
				v1 = lds_ld16(s + i + 16);
			if (rag)
				vt = lds_ld16(s + n - 16);
			lds_st16(d + i, v0);
			if (two)
				lds_st16(d + i + 16, v1);
			if (rag)
				lds_st16(d + n - 16, vt);
			if (last)
				break;
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

/* v >> (8 * bytes) over 128 bits, bytes >= 16 gives zeros (only the last 16 bytes of an image need it) */
__device__ __forceinline__ uint4 io_shr128(uint4 v, uint32_t bytes)
{
	uint32_t w[8] = { v.x, v.y, v.z, v.w, 0, 0, 0, 0 };
	for (; bytes >= 4 && bytes < 32; bytes -= 4) {
		w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = 0;
	}
	if (bytes >= 32)
		return make_uint4(0, 0, 0, 0);
	const uint32_t sh = 8 * bytes;
	if (sh) {
		w[0] = (w[0] >> sh) | (w[1] << (32 - sh));
		w[1] = (w[1] >> sh) | (w[2] << (32 - sh));
		w[2] = (w[2] >> sh) | (w[3] << (32 - sh));
		w[3] = w[3] >> sh;
	}
	return make_uint4(w[0], w[1], w[2], w[3]);
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

__global__ __launch_bounds__(IO_THREADS, 4) void lz4_expand_inorder_kernel(
    const uint8_t *__restrict__ src, uint64_t src_bytes, const la_lz4_block *__restrict__ blocks,
    uint32_t n, uint8_t *__restrict__ dst, uint64_t dst_cap, const uint64_t *__restrict__ dst_off,
    const uint32_t *__restrict__ out_len, uint32_t *status_out,
    const uint32_t *__restrict__ nseq, const la_lz4_seq *__restrict__ table,
    const uint64_t *__restrict__ table_off, uint32_t long_thr)
{
	const uint32_t *status = status_out;
	/* 16 bytes of headroom + up to 15 of alignment shift + the window + slack for over-reads */
	__shared__ __attribute__((aligned(16))) uint8_t win[16 + 16 + 65536 + 96];
	/* what L has worked out for M, one 16-byte record per sequence, IO_RING groups deep:
	 *   x = match destination (17 bits) | first << 17 | last << 23 | dep << 29: the lanes first..last of the group
	 *       are the ones whose matches the source of this match overlaps (dep = 0: none);
	 *   y = match length, z = match offset, w = running number of the group + 1 -- the tag that tells M the
	 *       record is there: a group's records are written by ONE ds_write_b128, after its literals. */
	__shared__ __attribute__((aligned(16))) uint4 ring[IO_RING][64];
	__shared__ io_ctl ctl_s;
	io_ctl *const ctl = &ctl_s;

	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0) {
		ctl->match_groups = 0;
		ctl->match_flag = 0x7FFFu << 17;	/* no block has this number first */
		ctl->flushed = 0;
		ctl->abort = 0;
	}
	for (uint32_t i = tid; i < IO_RING * 64u; i += IO_THREADS)
		ring[i >> 6][i & 63] = make_uint4(0, 0, 0, 0);	/* no group has tag 0 */
	__syncthreads();	/* the only barrier: the roles part here */

	const uint32_t b_first = blockIdx.x * IO_BPW;
	bool dead = false;	/* wave-uniform: this wave has seen a failure (its own or the abort word) */
	IO_STAMP_DECL;

#define IO_LOAD_BLK(bi_) io_load_blk((bi_), n, src, src_bytes, blocks, dst, dst_cap, dst_off, out_len, status, nseq, table, table_off, long_thr)
/* A wave that gives up marks the block and raises the abort word; it does NOT leave (hipcc 7.2 miscompiled the
 * literal waves' loop-carried prefetch registers around an early return inside the group loop: after the first
 * ring wait the next group's literal bytes were replaced by the current group's -- found with the reference's
 * test_compat_lz4_B4 fixture).  From then on every wait of every wave falls through at once, matches and literals
 * are skipped (dead), and the workgroup runs out quickly; the block's status tells the host. */
#define IO_FAIL(bi_)                                          \
	do {                                                  \
		if (lane == 0) {                              \
			status_out[bi_] = LA_ST_LZ4_DECODE;   \
			io_st(&ctl->abort, 1);                \
		}                                             \
		dead = true;                                  \
	} while (0)

	if (wave == 0) {
		/* ================= M: matches, in stream order ================= */
		__builtin_amdgcn_s_setprio(3);
#ifdef LA_DIAG
		io_acc[12] = __builtin_amdgcn_s_memrealtime();
#endif
		uint32_t gbase = 0, seqno = 0;
		uint4 xn = ring[0][lane];	/* the next group's records, requested a group ahead */
		for (uint32_t it = 0; it < IO_BPW; it++) {
			const uint32_t bi = b_first + it;
			const io_blk B = IO_LOAD_BLK(bi);
			if (!B.take)
				continue;
			uint8_t *const W = win + 16 + ((uintptr_t)B.g_out & 15);	/* W[i] <-> g_out[i], congruent mod 16 */
			const uint32_t w_lds = io_lds_addr(W);
			const uint32_t ng = (B.ns + 63) >> 6;
			[[maybe_unused]] const unsigned long long t_blk0 = IO_NOW();
			for (uint32_t g = 0; g < ng; g++) {
				const uint32_t G = gbase + g;
				/* L has to be past this group (it normally is, by several groups: then the records came with
				 * the previous group's loads).  All 64 records of a group appear at once. */
				[[maybe_unused]] const unsigned long long t_w0 = IO_NOW();
#ifdef IO_DBG_NOPREFETCH
				asm volatile("" ::: "memory");
				uint4 x = ring[G % IO_RING][lane];
#else
				uint4 x = xn;
#endif
				for (uint32_t spins = 0; __ballot(x.w == G + 1) != ~0ull;) {
					if (++spins > IO_SPIN_LIMIT || __builtin_amdgcn_readfirstlane((int)io_ld(&ctl->abort))) {
						IO_FAIL(bi);
						break;
					}
					asm volatile("" ::: "memory");
					x = ring[G % IO_RING][lane];
				}
				if (dead)
					x = make_uint4(0, 0, 0, 0);	/* nothing to copy */
				IO_STAMP_ADD(1, IO_NOW() - t_w0);
				asm volatile("" ::: "memory");
				const uint32_t mdst = x.x & 0x1FFFFu, mlen = x.y, off = x.z;
				const uint32_t d_lo = (x.x >> 17) & 63u, d_hi = (x.x >> 23) & 63u;
				const bool dep = (x.x >> 29) & 1u;
				/* the lanes this one has to wait for, as a mask */
#ifdef IO_DBG_NODEP	/* every lower lane counts as a dependency */
				const uint64_t rmask = (1ull << lane) - 1ull;
				(void)dep; (void)d_hi; (void)d_lo;
#else
				const uint64_t rmask = dep ? (2ull << d_hi) - (1ull << d_lo) : 0ull;
#endif
				bool fin = mlen == 0;
				uint8_t *const mp = W + mdst;
				const bool slow = mlen > 64 || off < mlen;
				const uint32_t mf = (mlen != 0 && !slow) ? mlen : 0u;	/* length if the straight-line copy takes the match */
				io_cls K;
				K.k16 = __ballot(mf >= 16);
				K.k32 = __ballot(mf >= 32);
				K.k48 = __ballot(mf >= 48);
				K.ke = __ballot(mf > 16 && mf != 32 && mf != 48);
				K.ks = __ballot(mf != 0 && mf < 16);
				K.ks8 = __ballot(mf >= 8 && mf < 16);
				K.ks88 = __ballot(mf > 8 && mf < 16);
				K.ks4 = __ballot(mf >= 4 && mf < 8);
				K.ks44 = __ballot(mf > 4 && mf < 8);
				K.k2 = __ballot(mf != 0 && mf < 4 && (mf & 2));
				K.k1 = __ballot(mf != 0 && mf < 4 && (mf & 1));
				const uint64_t kslow = __ballot(slow && mlen != 0);
				io_adr A;
				A.mp = w_lds + mdst;
				A.fp = A.mp - off;
				A.me = A.mp + mlen - 16;
				A.fe = A.fp + mlen - 16;
				A.m8 = A.mp + mlen - 8;
				A.f8 = A.fp + mlen - 8;
				A.m4 = A.mp + mlen - 4;
				A.sh4 = 8 * (mlen - 4);
				A.m1 = A.mp + (mlen & 2);
				A.sh1 = 8 * (mlen & 2);
				/* the next group's records travel with this group's first loads */
				asm volatile("" ::: "memory");
				xn = ring[(G + 1) % IO_RING][lane];
				asm volatile("" ::: "memory");
				[[maybe_unused]] const unsigned long long t_p0 = IO_NOW();
				[[maybe_unused]] unsigned long long t_p1 = t_p0;

				/* pass 1: every match whose source does not reach into a match of its own group.  Then rounds:
				 * a lane is ready when none of the lanes it waits for is unfinished (the lowest unfinished
				 * lane always is) */
				uint64_t unf = __ballot(!fin);
				uint32_t rounds = 0;
				while (unf) {
					const bool ready = !fin && (rmask & unf) == 0;
					const uint64_t R = __ballot(ready);
#ifndef IO_EXP_NO_COPY	/* timing experiment */
					io_copy_fast(R, K, A);
#endif
					if (R & kslow) {	/* longer than 64 bytes or overlapping its source: rare */
						if (ready && slow)
							io_copy_match(mp, off, mlen);
					}
					fin = fin || ready;
					asm volatile("" ::: "memory");
#ifdef LA_DIAG
					if (rounds == 0) t_p1 = IO_NOW();
#endif
#ifdef IO_EXP_NO_LATE	/* timing experiment: one pass only (wrong output) */
					break;
#endif
					unf &= ~R;
					if (R == 0 || ++rounds > 64u) {	/* cannot happen: the lowest unfinished lane is always ready */
						IO_FAIL(bi);
						break;
					}
				}
				IO_STAMP_ADD(2, rounds);
#ifdef LA_DIAG
				IO_STAMP_ADD(6, t_p1 - t_p0);
				IO_STAMP_ADD(7, IO_NOW() - t_p1);
#endif
				/* everything below the end of the group's last sequence is final */
				const uint32_t s_next = (uint32_t)__builtin_amdgcn_readlane((int)(mdst + mlen), 63);
				asm volatile("" ::: "memory");
				if (lane == 0) {
					io_st(&ctl->match_flag, ((seqno & 0x7FFFu) << 17) | s_next);
					io_st(&ctl->match_groups, G + 1);
				}
			}
			IO_STAMP_ADD(0, IO_NOW() - t_blk0);
			IO_STAMP_ADD(3, ng);
			gbase += ng;
			seqno++;
		}
#ifdef LA_DIAG
		io_acc[13] = __builtin_amdgcn_s_memrealtime();
#endif
		IO_STAMP_FLUSH();
	} else if (wave <= IO_LWAVES) {
		/* ================= L: table + literals, several groups ahead of M =================
		 * IO_LWAVES waves share the groups out by their running number G (wave j takes G % IO_LWAVES == j);
		 * each keeps the entries two of its groups ahead and the literal bytes one ahead, every load
		 * unconditional (indices and addresses are clamped into the table / the image, not predicated):
		 * one global round trip per iteration and wave, IO_LWAVES groups per round trip.  Per group: literals
		 * into the window, then what M needs to know about the matches (where, how long, from where, and which
		 * matches of the same group each one has to wait for) as one record per sequence. */
		const uint32_t lw = wave - 1;
		uint32_t gbase = 0, seqno = 0;
		for (uint32_t it = 0; it < IO_BPW; it++) {
			const uint32_t bi = b_first + it;
			const io_blk B = IO_LOAD_BLK(bi);
			if (!B.take)
				continue;
			uint8_t *const W = win + 16 + ((uintptr_t)B.g_out & 15);
			const uint32_t ns = B.ns, olen = B.olen;
			const uint32_t ng = (ns + 63) >> 6;
			const uint8_t *const s = B.s;
			const uint64_t s_room = B.s_room;
			/* loads of 16 payload bytes never leave the image: an address closer than 16 bytes to its
			 * end is pulled back (dmax may be negative: the image is at least 16 bytes, the launcher sees
			 * to that) and the bytes are shifted down when they are used */
			const int64_t dm64 = (int64_t)s_room - 16;
			const int32_t dmax = dm64 > 0x7FFFFFFFll ? 0x7FFFFFFF : (int32_t)dm64;
			const uint64_t *const tab64 = (const uint64_t *)(const void *)B.tab;
#define IO_LOAD_E(grp_) tab64[((grp_) * 64 + lane) < ns ? ((grp_) * 64 + lane) : ns - 1]
#define IO_LOAD_N(grp_) tab64[((grp_) + 1) * 64 < ns ? ((grp_) + 1) * 64 : ns - 1]	/* first entry of the group behind it */
#define IO_LOAD_P(p0_, pt_, sh0_, sht_, e_)                                                               \
	do {                                                                                              \
		const uint32_t ls_ = SEQ_LIT_SRC(e_), ll_ = SEQ_LIT_LEN(e_);                              \
		const int32_t a0_ = (int32_t)ls_ < dmax ? (int32_t)ls_ : dmax;                            \
		const uint32_t tl_ = ll_ > 32u ? ls_ + 16u : (ll_ > 16u ? ls_ + ll_ - 16u : ls_);         \
		const int32_t at_ = (int32_t)tl_ < dmax ? (int32_t)tl_ : dmax;                            \
		p0_ = ld_u128(s + a0_);                                                                   \
		pt_ = ld_u128(s + at_);                                                                   \
		sh0_ = ls_ - (uint32_t)a0_;                                                               \
		sht_ = tl_ - (uint32_t)at_;                                                               \
	} while (0)
			const uint32_t g0 = (lw + IO_LWAVES - (gbase % IO_LWAVES)) % IO_LWAVES;	/* this wave's first group of the block */
			/* prologue: the first groups' entries and literal bytes are requested before the window is
			 * even free */
			seq_t eA = IO_LOAD_E(g0), nA = IO_LOAD_N(g0);
			seq_t eB = IO_LOAD_E(g0 + IO_LWAVES), nB = IO_LOAD_N(g0 + IO_LWAVES);
			uint4 pA0, pAt;
			uint32_t shA0, shAt;
			IO_LOAD_P(pA0, pAt, shA0, shAt, eA);
			/* the window is ours when F has read the previous block out */
			[[maybe_unused]] const unsigned long long t_w0 = IO_NOW();
			if (!io_wait_ge(&ctl->flushed, seqno, ctl, 1))
				IO_FAIL(bi);
			IO_STAMP_ADD(5, IO_NOW() - t_w0);
			asm volatile("" ::: "memory");
			[[maybe_unused]] const unsigned long long t_l0 = IO_NOW();
			for (uint32_t g = g0; g < ng; g += IO_LWAVES) {
				/* entries two of this wave's groups ahead, literal bytes one ahead */
				const seq_t eC = IO_LOAD_E(g + 2 * IO_LWAVES), nC = IO_LOAD_N(g + 2 * IO_LWAVES);
				uint4 pB0, pBt;
				uint32_t shB0, shBt;
				IO_LOAD_P(pB0, pBt, shB0, shBt, eB);
				/* this group */
				const seq_t e = eA;
				const uint32_t k = g * 64 + lane;
				const bool valid = k < ns;
				/* where the next group's output begins = where this group's ends */
				const uint32_t s_next = (g + 1) * 64 < ns ? (uint32_t)__builtin_amdgcn_readfirstlane((int)SEQ_DST(nA)) : olen;
				const uint32_t d = valid ? SEQ_DST(e) : olen;
				const uint32_t ls = SEQ_LIT_SRC(e), off = SEQ_OFF(e);
				uint32_t ll = valid ? SEQ_LIT_LEN(e) : 0u;
				const uint32_t mdst = d + ll;
				/* output position of the next sequence: lane + 1's, the next group's first for lane 63 */
				uint32_t nd = (uint32_t)__builtin_amdgcn_update_dpp((int)s_next, (int)d, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
				if (k + 1 >= ns)
					nd = olen;	/* the last sequence ends with the block */
				/* an entry that does not add up (impossible for a table the parse kernel wrote) fails the block */
				const bool bad = valid && (mdst > olen || nd < mdst || nd > olen || (uint64_t)ls + ll > s_room ||
				    (nd != mdst && (off == 0 || off > mdst)));
				if (__ballot(bad) != 0)
					IO_FAIL(bi);
				const uint32_t mlen = (valid && !dead) ? nd - mdst : 0u;
				if (dead)
					ll = 0;		/* (mdst stays as computed: only consistent positions are handed on) */
				const uint32_t G = gbase + g;
#ifdef IO_EXP_NO_LIT	/* timing experiment: no literal stores (wrong output) */
				if (ll > 0x100000u) {
#else
				if (ll) {
#endif
					uint8_t *wd = W + d;
					uint4 p0 = pA0, pt = pAt;
#ifdef IO_DBG_SYNC_P
					IO_LOAD_P(p0, pt, shA0, shAt, e);
#endif
#ifdef IO_DBG_NOSH
					if (0) {
#else
					if (shA0 | shAt) {	/* inside the last 16 bytes of the image */
#endif
						p0 = io_shr128(p0, shA0);
						pt = io_shr128(pt, shAt);
					}
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
				/* Which matches of this group does the source of each match overlap?  The matches of the group lie in
				 * stream order: [bm, em) of lane i ends before lane i + 1's begins.  The source [s0, send) (send: the
				 * end of the part that exists before the copy starts) meets the lanes first .. last with
				 * first = #{i: em_i <= s0}, last = #{i: bm_i < send} - 1: two binary searches over the lanes
				 * (ds_bpermute), run side by side.  Both counts are below this lane's own number. */
				const uint32_t bm = valid ? mdst : olen, em = bm + mlen;
				const uint32_t s0 = bm - off, send = s0 + (mlen < off ? mlen : off);
				uint32_t cl = 0, ch = 0;
#pragma unroll
				for (uint32_t step = 32; step; step >>= 1) {
					const uint32_t ve = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((cl + step - 1) << 2), (int)em);
					const uint32_t vb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((ch + step - 1) << 2), (int)bm);
					if (ve <= s0) cl += step;
					if (vb < send) ch += step;
				}
				const bool dep = mlen != 0 && cl < ch;	/* first = cl <= last = ch - 1 */
				/* the ring slot is free when M is done with the group IO_RING places back */
				[[maybe_unused]] const unsigned long long t_r0 = IO_NOW();
				if (G >= IO_RING && !io_wait_ge(&ctl->match_groups, G + 1 - IO_RING, ctl, 1))
					IO_FAIL(bi);
				IO_STAMP_ADD(11, IO_NOW() - t_r0);
				/* the records go out last, in one instruction: their tag tells M that the group's literals are in
				 * the window too (the LDS runs a wave's instructions in order) */
#ifdef IO_DBG_LWAIT
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
				asm volatile("" ::: "memory");
				ring[G % IO_RING][lane] = make_uint4(bm | (dep ? (cl << 17) | ((ch - 1) << 23) | (1u << 29) : 0u), mlen, valid ? off : 0u, G + 1);
				asm volatile("" ::: "memory");
				eA = eB; nA = nB;
				eB = eC; nB = nC;
				pA0 = pB0; pAt = pBt; shA0 = shB0; shAt = shBt;
			}
#undef IO_LOAD_E
#undef IO_LOAD_N
#undef IO_LOAD_P
			IO_STAMP_ADD(8, IO_NOW() - t_l0);
			gbase += ng;
			seqno++;
		}
		IO_STAMP_FLUSH();
	} else if (wave == IO_LWAVES + 1) {
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
#ifdef IO_EXP_NO_FLUSH	/* timing experiment: nothing is written out */
				done = avail;
#endif
				while (avail - done >= 256u) {	/* 4 KiB at a time: four LDS reads in flight, then four stores */
					const uint32_t u = done + lane;
					const uint4 r0 = wsrc[u], r1 = wsrc[u + 64], r2 = wsrc[u + 128], r3 = wsrc[u + 192];
					gdst[u] = r0;
					gdst[u + 64] = r1;
					gdst[u + 128] = r2;
					gdst[u + 192] = r3;
					done += 256u;
					moved = true;
				}
				while (complete && done < avail) {	/* the last, partial 4 KiB */
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
				} else if (++spins > IO_SPIN_LIMIT || __builtin_amdgcn_readfirstlane((int)io_ld(&ctl->abort))) {
					IO_FAIL(bi);
					break;
				}
				IO_STAMP_ADD(9, 1);
				__builtin_amdgcn_s_sleep(4);
			}
			/* every LDS read of this block has returned (its data went into the stores above) */
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			seqno++;
			if (lane == 0)
				io_st(&ctl->flushed, seqno);
		}
		IO_STAMP_FLUSH();
	}
#undef IO_LOAD_BLK
#undef IO_FAIL
}

#ifdef LA_DIAG
extern "C" int la_diag_set_io_stamps(void *d_buf)
{
	unsigned long long *p = (unsigned long long *)d_buf;
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(la_diag_io_stamps), &p, sizeof(p));
}
#endif

/* the kernel's payload loads are pulled back from the end of the image by up to 16 bytes: images shorter than
 * that go to the previous-generation kernel (the caller's choice, la_api.hip) */
bool la_lz4_expand_inorder_takes(uint64_t src_bytes) { return src_bytes >= 16; }

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
