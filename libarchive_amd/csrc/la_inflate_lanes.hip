/*
 * la_inflate_lanes.hip -- raw DEFLATE decode, ONE LANE PER MEMBER (gfx950).
 *
 * Same job and same accept/reject rules as la_inflate.hip (zlib inflate(),
 * gzip.c:431-511 / :479; rules in oracle/orc_inflate.c), different shape: a
 * deflate stream is one serial bit chain, so for batches of MANY independent
 * members (BGZF-style multi-member streams, SURVEY config 3: 262 144 members)
 * the parallel axis is "members", exactly as the LZ4 parse kernel walks one token
 * chain per lane.  Each lane owns
 *   - a 64-bit bit buffer refilled with unaligned 8-byte loads of its member,
 *   - its Huffman code in REGISTERS and LDS (round 3, IL_CANON): the literal/length code is resolved by a
 *     canonical walk -- the next 15 bits, left-justified, against one limit per code length (the limits grow
 *     with the length), branch-free, four operations per length; limit, index bias and the "symbols from 256
 *     up start here" mark of every length live in registers.  LDS holds only what needs an index:
 *         288 x u8  the literal/length symbols in canonical order (low byte), TRANSPOSED ([entry][lane]) so
 *                   that 64 lanes looking up 64 different symbols do not queue on one lane's row,
 *         32 x u8   distance fast table (5-bit index; longer distance codes: a walk over counts, the at most
 *                   32 symbols sit in registers), also used for the code-length code of a block header,
 *     = 320 B per lane, 80 KiB per 256-lane workgroup: TWO workgroups per CU, two waves per SIMD,
 *   - its output slot in the slab: literals are single byte stores, matches are
 *     8-byte "wild" copies (a lane reads back its own earlier stores, which the
 *     hardware keeps coherent per thread; bytes past a match are overwritten by
 *     what follows, and never lie beyond the member's slot).
 * The wave-per-member kernel of la_inflate.hip stays the path for small batches
 * and for members whose slot is too small for wild copies.
 *
 * How it got here (entropy decode of the 16 GiB C3 stream, ms; profiles/README.md round 3):
 *   94.1  round 2: 7-bit literal/length fast table + 6-bit distance table + symbol list, 608 B per lane, ONE
 *         workgroup per CU.  Two workgroups per CU were tried then with smaller tables or the symbol list in
 *         global memory and were SLOWER (114-144): the measurement was dominated by something else --
 *   81.2  the ISA showed every bit-buffer refill waiting for its OWN prefetch (s_waitcnt vmcnt(0) in front of a
 *         register copy at the end of `if (take) {...}`: a phi of the old and the new load) and every long
 *         distance code fetching its symbol from global scratch; SQ_WAIT_ANY was 52 % of the wave cycles.
 *         Branch-free refill, stores collected at flush points right behind the refill, distance symbols in
 *         registers: 38 % waiting, the rest is instruction issue -- one wave per SIMD issues a dependent
 *         instruction only every ~8 cycles.
 *   46.7  so a second wave per SIMD pays once the tables fit twice: on the C3 members nearly every literal code
 *         is longer than the 7-bit table anyway (random literals: 8-9 bits), the table bought nothing there, and
 *         the canonical walk costs the same whatever the code length.  (IL_CANON=0 keeps the table version.)
 */
#include "la_dev.h"

#ifndef IL_THREADS
#define IL_THREADS 256
#endif
#ifndef IL_CANON
#define IL_CANON 1	/* 1: no literal/length fast table; every literal/length code resolves by the canonical walk over
			 * per-length limits kept in registers, 320 B of LDS per lane, TWO workgroups per CU */
#endif
#ifndef LL_BITS
#define LL_BITS 7
#endif
#ifndef DT_BITS
#define DT_BITS (IL_CANON ? 5 : 6)
#endif
#ifndef IL_SL_GLOBAL
#define IL_SL_GLOBAL 0	/* 1: the literal/length symbol list for long codes lives in global scratch, not LDS */
#endif
#ifndef IL_LIT_ROUNDS
#define IL_LIT_ROUNDS 2
#endif
#ifndef IL_LIT_BURST
#define IL_LIT_BURST 3	/* (3 x 15 bits fit the one refill of a burst) */
#endif
/* Diagnostic build only (make diag): wave-level counters, added by the first active lane of whatever
 * subset of the wave runs the region.  No output value depends on them. */
#ifdef LA_DIAG
__device__ unsigned long long *la_diag_il;
#define IL_FIRST_LANE() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == __builtin_ffsll((long long)__ballot(1)) - 1)
#define IL_CNT(slot, v) do { if (la_diag_il && IL_FIRST_LANE()) atomicAdd(&la_diag_il[slot], (unsigned long long)(v)); } while (0)
#define IL_NOW() __builtin_readcyclecounter()
#else
#define IL_CNT(slot, v) do { } while (0)
#define IL_NOW() 0ull
#endif
#define IL_SCRATCH_PER_LANE 1024u	/* bytes of global scratch per member: lens[320] + sorted symbols u16[320] + pad */

/* ---- per-lane LDS tables, transposed: element e of lane t at [e][t] ---- */
/* 608 bytes per lane = 152 KiB per 256-lane workgroup (one workgroup per CU):
 *   ll  fast table of the literal/length code, LL_BITS wide, entry = symbol << 4 | length
 *   sl  the literal/length symbols in canonical code order (low byte; symbols above 255 come
 *       last within their length, the lane keeps that boundary per length in registers), so
 *       a code longer than the fast table still resolves without a trip to global memory --
 *       every lane of the wave waits for such a trip whenever one lane needs it, and with
 *       64 lanes one nearly always does
 *   dt  fast table of the distance code (and, while a block header is read, of the
 *       code-length code), DT_BITS wide, entry = symbol << 3 | length */
struct il_lds {
#if !IL_CANON
	uint16_t ll[1 << LL_BITS][IL_THREADS];
#endif
#if !IL_SL_GLOBAL
	uint8_t sl[288][IL_THREADS];
#endif
	uint8_t dt[1 << DT_BITS][IL_THREADS];
};

/* 16 counters of up to 15 bits each, packed 4 per u64 (dynamic index without scratch) */
struct packed16 {
	uint64_t w[4];
};
__device__ __forceinline__ uint32_t p16_get(const packed16 &p, uint32_t i)
{
	const uint64_t v = (i >> 2) == 0 ? p.w[0] : (i >> 2) == 1 ? p.w[1] : (i >> 2) == 2 ? p.w[2] : p.w[3];
	return (uint32_t)(v >> (16 * (i & 3))) & 0xFFFFu;
}
__device__ __forceinline__ void p16_add(packed16 &p, uint32_t i, uint32_t add)
{
	const uint64_t a = (uint64_t)add << (16 * (i & 3));
	if ((i >> 2) == 0) p.w[0] += a; else if ((i >> 2) == 1) p.w[1] += a;
	else if ((i >> 2) == 2) p.w[2] += a; else p.w[3] += a;
}
__device__ __forceinline__ void p16_zero(packed16 &p) { p.w[0] = p.w[1] = p.w[2] = p.w[3] = 0; }

struct lane_bits {
	const uint8_t *s;	/* member's first deflate byte */
	uint64_t hold;
	uint64_t cur, nxt;	/* the 8 bytes at ip and at ip+8: nxt is requested one refill ahead */
	uint32_t bits;		/* valid bits in hold */
	uint32_t ip;		/* next byte to feed into hold */
	uint32_t iend;		/* bytes of the member's span */
	uint32_t room;		/* bytes from s to the end of the source image */
	const uint8_t *safe;	/* any address with 8 readable bytes (used instead of an out-of-image one) */
	uint32_t nxt_drop;	/* leading bytes of nxt to shift out when it is used */
};

/* Eight input bytes from offset `at`, zeros past the end of the image.  Branch-free in the
 * common case ON PURPOSE: with a branch around the load the compiler has to merge the loaded
 * value with the slow path's at the join, which puts the wait for the load right behind it --
 * and the whole point of `nxt` is that nobody waits for it until the next refill.  The address
 * is clamped so that the load never leaves the image; what was clamped away is shifted out. */
__device__ __forceinline__ uint64_t lb_load8_raw(const lane_bits &B, uint32_t at, uint32_t *drop)
{
	const bool tiny = B.room < 8;	/* (an image tail shorter than one load: bytes one by one) */
	const uint32_t a2 = tiny ? 0u : (at < B.room - 8u ? at : B.room - 8u);
	const uint8_t *ptr = tiny ? B.safe : B.s + a2;
	uint64_t v;
	__builtin_memcpy(&v, ptr, 8);
	*drop = at - a2;	/* bytes of v in front of `at`: shifted out when v is USED, not here --
				 * touching v now would be waiting for the load now */
	if (tiny) {
		v = 0;
		for (uint32_t k = 0; k < 8; k++)
			if (at + k < B.room)
				v |= (uint64_t)B.s[at + k] << (8 * k);
		*drop = 0;
	}
	return v;
}
__device__ __forceinline__ uint64_t lb_fix(uint64_t raw, uint32_t drop) { return drop >= 8 ? 0 : raw >> (8 * drop); }

__device__ __forceinline__ void lb_start(lane_bits &B)
{
	B.hold = 0; B.bits = 0; B.ip = 0;
	uint32_t d0;
	B.cur = lb_load8_raw(B, 0, &d0);
	B.cur = lb_fix(B.cur, d0);
	B.nxt = lb_load8_raw(B, 8, &B.nxt_drop);
}

/* re-aim the byte window after the stored-block path moved ip by hand */
__device__ __forceinline__ void lb_seek(lane_bits &B, uint32_t ip)
{
	B.ip = ip; B.bits = 0; B.hold = 0;
	uint32_t d0;
	B.cur = lb_load8_raw(B, ip, &d0);
	B.cur = lb_fix(B.cur, d0);
	B.nxt = lb_load8_raw(B, ip + 8, &B.nxt_drop);
}

/* top the buffer up to >= 56 bits.  The bytes come from `cur`; the load that replaces
 * what was used is issued now and only needed at the NEXT refill, so its latency is
 * hidden behind the symbols in between.  Bytes past iend may come along (they exist in
 * the image or are zero) and are never COUNTED as available: see lb_avail().
 * NO BRANCH on purpose (round 3): with `if (take)` around the reload the old and the new `nxt`
 * meet in a phi, the compiler copies the freshly loaded register pair at the end of the branch
 * and puts `s_waitcnt vmcnt(0)` in front of the copy -- every refill then waited for its OWN
 * prefetch, a full global round trip, two to three times per outer iteration (seen in the ISA;
 * 52 % of the wave cycles were SQ_WAIT_ANY).  With take == 0 the same address is simply asked
 * for again. */
__device__ __forceinline__ void lb_refill(lane_bits &B)
{
	B.hold |= B.cur << B.bits;
	const uint32_t take = (63u - B.bits) >> 3;
	B.bits += take * 8;
	B.ip += take;
	const uint64_t nx = lb_fix(B.nxt, B.nxt_drop);	/* requested one refill ago */
	const uint64_t moved = (B.cur >> (8 * take)) | (nx << ((64 - 8 * take) & 63));
	B.cur = take ? moved : B.cur;
	B.nxt = lb_load8_raw(B, B.ip + 8, &B.nxt_drop);
}
/* bits of REAL input still unread (may be negative when the buffer ran past iend) */
__device__ __forceinline__ int32_t lb_avail(const lane_bits &B)
{
	return (int32_t)B.bits - (int32_t)((int64_t)B.ip - (int64_t)B.iend) * 8 * (B.ip > B.iend ? 1 : 0);
}
__device__ __forceinline__ uint32_t lb_peek(const lane_bits &B, uint32_t n) { return (uint32_t)B.hold & ((1u << n) - 1u); }
__device__ __forceinline__ void lb_drop(lane_bits &B, uint32_t n) { B.hold >>= n; B.bits -= n; }

/* per-lane canonical code description kept in registers for the slow path */
struct lane_code {
	packed16 count;		/* codes per length 1..15 */
	uint32_t maxlen;
	uint32_t first_p, index_p;	/* canonical-walk state after the lengths the fast table covers */
	packed16 nlow;		/* literal/length code only: symbols below 256 per length */
#if IL_CANON
	/* literal/length code, canonical walk: per length k, lim[k] = (left-justified 15-bit end of the codes of
	 * length k) | (index of the length's first symbol - its first code) << 16, hib[k] = index from which the
	 * length's symbols are 256 and up; bit k of `has`: the code has words of length k */
	uint32_t lim[16], hib[16], has;
#endif
	uint32_t y0, y1, y2, y3, y4, y5, y6, y7;	/* distance / code-length code only: the (at most 32) symbols in
				 * canonical order, one byte each, IN REGISTERS (named scalars: an array
				 * would be indexed through scratch memory): a long distance code then
				 * costs a few selects instead of a trip to global scratch that the whole
				 * wave waits for */
};
__device__ __forceinline__ void lc_sym_set(lane_code &C, uint32_t pos, uint32_t sy)
{
	const uint32_t sh = 8u * (pos & 3u), w = pos >> 2, keep = ~(0xFFu << sh), put = sy << sh;
#define LC_SET(i, f) C.f = w == i ? ((C.f & keep) | put) : C.f
	LC_SET(0, y0); LC_SET(1, y1); LC_SET(2, y2); LC_SET(3, y3);
	LC_SET(4, y4); LC_SET(5, y5); LC_SET(6, y6); LC_SET(7, y7);
#undef LC_SET
}
/* (masks, not selects: a chain of selects over loads from the struct is folded by the compiler into ONE load
 * through a selected address, which pins the whole struct in scratch memory) */
__device__ __forceinline__ uint32_t lc_sym_get(const lane_code &C, uint32_t pos)
{
	const uint32_t w = pos >> 2;
#define LC_M(i) (0u - (uint32_t)(w == i))
	const uint32_t v = (C.y0 & LC_M(0)) | (C.y1 & LC_M(1)) | (C.y2 & LC_M(2)) | (C.y3 & LC_M(3)) |
	    (C.y4 & LC_M(4)) | (C.y5 & LC_M(5)) | (C.y6 & LC_M(6)) | (C.y7 & LC_M(7));
#undef LC_M
	return (v >> (8u * (pos & 3u))) & 0xFFu;
}

/*
 * Build one table from lens[0..n) (global scratch, this lane's): counts, sorted symbol
 * list (global scratch, for codes longer than the fast table) and the LDS fast table.
 * Returns 0 complete, >0 incomplete, <0 over-subscribed.
 */
template <int FAST_BITS, bool LL>
__device__ __forceinline__ int il_build(const uint8_t *lens, int n, lane_code &C, uint16_t *sorted,
    il_lds &T, int tid)
{
	p16_zero(C.count);
	p16_zero(C.nlow);
	if (!LL)
		C.y0 = C.y1 = C.y2 = C.y3 = C.y4 = C.y5 = C.y6 = C.y7 = 0;
	for (int i = 0; i < n; i++)
		p16_add(C.count, lens[i], 1);
	int left = 1, maxlen = 0;
	packed16 next_off, next_code;
	p16_zero(next_off); p16_zero(next_code);
	uint32_t off = 0, code = 0;
	for (int l = 1; l < 16; l++) {
		const uint32_t c = p16_get(C.count, l);
		if (c) maxlen = l;
		left = left * 2 - (int)c;
		if (left < 0) left = -100000;
		p16_add(next_off, l, off);
		off += c;
		code = (code + (l > 1 ? p16_get(C.count, l - 1) : 0)) << 1;
		p16_add(next_code, l, code & 0x7FFFu);
	}
	C.maxlen = (uint32_t)maxlen;
	{
		uint32_t fi = 0, ix = 0;
		for (int l = 1; l <= FAST_BITS; l++) {
			const uint32_t c = p16_get(C.count, l);
			ix += c;
			fi = (fi + c) << 1;
		}
		C.first_p = fi;
		C.index_p = ix;
	}
#if IL_CANON
	if (LL) {
		uint32_t cd = 0, ix = 0;
		C.has = 0;
		C.lim[0] = C.hib[0] = 0;
#pragma unroll
		for (int l = 1; l < 16; l++) {
			const uint32_t c = p16_get(C.count, l);
			cd = (cd + (l > 1 ? p16_get(C.count, l - 1) : 0u)) << 1;	/* first code of length l */
			C.lim[l] = (((cd + c) << (15 - l)) & 0xFFFFu) | ((ix - cd) << 16);
			C.hib[l] = ix;	/* (+ the literals of this length, below) */
			C.has |= (c ? 1u : 0u) << l;
			ix += c;
		}
	} else
#endif
	for (int i = 0; i < (1 << FAST_BITS); i++) {
#if !IL_CANON
		if (LL) T.ll[i][tid] = 0; else
#endif
		T.dt[i][tid] = 0;
	}
	if (left < 0)
		return -1;
	for (int sy = 0; sy < n; sy++) {
		const uint32_t l = lens[sy];
		if (l == 0) continue;
		const uint32_t pos = p16_get(next_off, l);
		p16_add(next_off, l, 1);
		if (LL && !IL_SL_GLOBAL) {
#if !IL_SL_GLOBAL
			T.sl[pos][tid] = (uint8_t)sy;
#endif
			if (sy < 256)
				p16_add(C.nlow, l, 1);
		} else if (LL) {
			sorted[pos] = (uint16_t)sy;
		} else {
			lc_sym_set(C, pos & 31u, (uint32_t)sy);
		}
		const uint32_t cw = p16_get(next_code, l);
		p16_add(next_code, l, 1);
		if (l <= (uint32_t)FAST_BITS && !(IL_CANON && LL)) {
			const uint32_t r = __builtin_bitreverse32(cw) >> (32 - l);
			for (uint32_t idx = r; idx < (1u << FAST_BITS); idx += (1u << l)) {
#if !IL_CANON
				if (LL) T.ll[idx][tid] = (uint16_t)((sy << 4) | l);
				else
#endif
				T.dt[idx][tid] = (uint8_t)((sy << 3) | l);
			}
		}
	}
#if IL_CANON
	if (LL) {
#pragma unroll
		for (int l = 1; l < 16; l++)
			C.hib[l] += p16_get(C.nlow, l);
	}
#endif
	return left;
}

/* decode one symbol; *used = bits consumed.  >= 0 symbol, -2 unassigned code.
 * The caller checks availability of the consumed bits afterwards. */
template <int FAST_BITS, bool LL>
__device__ __forceinline__ int il_decode(lane_bits &B, const lane_code &C, const uint16_t *sorted,
    const il_lds &T, int tid, uint32_t *used, uint32_t whas = 0)
{
#if IL_CANON
	if (LL) {
		/* the code's 15-bit left-justified value against the per-length limits (they grow with the
		 * length): the first limit above it is the code's length.  Lengths nobody in the wave has words
		 * of cost a test and a jump; a length somebody has costs seven operations for everybody. */
		const uint32_t rev = __builtin_bitreverse32(lb_peek(B, 15)) >> 17;	/* first bit read = bit 14 */
		IL_CNT(4, 1);
		/* Branch-free, longest length first: every length whose limit lies above the code overwrites the
		 * choice, so the SHORTEST such length stands at the end (the limits grow with the length; lengths
		 * the code has no words of repeat their neighbour's limit and change nothing).  Four operations per
		 * length; the rare lengths (1-3 and 13-15 bits) sit behind one wave-uniform test each (`whas`:
		 * the lengths some lane of the wave has words of, set up once per block). */
		uint32_t hit_len = 0, bias = 0, hb = 0;
#define IL_TRY(k)                                                                     \
		do {                                                                  \
			const uint32_t a_ = C.lim[k];                                 \
			const bool under_ = rev < (a_ & 0xFFFFu);                     \
			hit_len = under_ ? (uint32_t)(k) : hit_len;                   \
			bias = under_ ? (uint32_t)((int32_t)a_ >> 16) : bias;         \
			hb = under_ ? C.hib[k] : hb;                                  \
		} while (0)
		if (whas & 0xE000u) { IL_TRY(15); IL_TRY(14); IL_TRY(13); }
		IL_TRY(12); IL_TRY(11); IL_TRY(10); IL_TRY(9); IL_TRY(8); IL_TRY(7); IL_TRY(6); IL_TRY(5); IL_TRY(4);
		if (whas & 0x000Eu) { IL_TRY(3); IL_TRY(2); IL_TRY(1); }
#undef IL_TRY
		IL_CNT(5, 9);
		const uint32_t hit_idx = (rev >> (15u - hit_len)) + bias;	/* (hit_len 0: unused) */
		const uint32_t hit_hi = hit_idx >= hb ? 256u : 0u;
		if (hit_len) {
			lb_drop(B, hit_len);
			*used = hit_len;
			return (int)((uint32_t)T.sl[hit_idx][tid] | hit_hi);
		}
		const uint32_t ml = C.maxlen ? C.maxlen : 1;
		lb_drop(B, ml);
		*used = ml;
		return -2;
	}
	const uint32_t e = (uint32_t)T.dt[lb_peek(B, FAST_BITS)][tid];
	const uint32_t l = e & 7u;
#else
	const uint32_t e = LL ? (uint32_t)T.ll[lb_peek(B, FAST_BITS)][tid] : (uint32_t)T.dt[lb_peek(B, FAST_BITS)][tid];
	const uint32_t l = LL ? (e & 15u) : (e & 7u);
#endif
	if (l) {
		lb_drop(B, l);
		*used = l;
		return (int)(LL ? (e >> 4) : (e >> 3));
	}
	/* Long or unassigned code.  The canonical walk for the lengths the fast table covers
	 * cannot hit (the table would have had the code) and its state after them does not
	 * depend on the code: it was precomputed with the table.  The remaining lengths are
	 * tried in straight-line code on the next 15 bits (every lane of the wave pays for this
	 * path whenever one lane takes it, so it has no loop and no dynamic counter picks). */
	const uint32_t rev = __builtin_bitreverse32(lb_peek(B, 15)) >> 17;	/* first bit read = bit 14 */
	IL_CNT(LL ? 4 : 8, 1);
	int first = (int)C.first_p, index = (int)C.index_p;
	int hit_len = 0, hit_idx = 0, hit_hi = 0;
#pragma unroll
	for (int k = FAST_BITS + 1; k <= 15; k++) {
		if (__ballot(hit_len == 0 && (uint32_t)k <= C.maxlen) == 0)
			break;	/* (wave-uniform) every lane here has its code, or no longer ones exist */
		IL_CNT(LL ? 5 : 9, 1);
		const int cn = (int)((C.count.w[k >> 2] >> (16 * (k & 3))) & 0xFFFFu);
		const int codev = (int)(rev >> (15 - k));
		const bool hit = hit_len == 0 && (uint32_t)k <= C.maxlen && codev - cn < first;
		hit_idx = hit ? index + (codev - first) : hit_idx;
		if (LL)	/* symbols above 255 are the last ones of their length */
			hit_hi = hit ? ((codev - first) >= (int)((C.nlow.w[k >> 2] >> (16 * (k & 3))) & 0xFFFFu) ? 256 : 0) : hit_hi;
		hit_len = hit ? k : hit_len;
		index += cn;
		first = (first + cn) << 1;
	}
	if (hit_len) {
		lb_drop(B, (uint32_t)hit_len);
		*used = (uint32_t)hit_len;
#if !IL_SL_GLOBAL
		if (LL)
			return (int)T.sl[hit_idx][tid] | hit_hi;
#endif
		if (!LL)
			return (int)lc_sym_get(C, (uint32_t)hit_idx & 31u);
		return sorted[hit_idx];
	}
	{
		const uint32_t ml = C.maxlen ? C.maxlen : 1;
		lb_drop(B, ml);
		*used = ml;
	}
	return -2;
}

/* length / distance symbol -> base value and extra bits, by arithmetic (RFC 1951 3.2.5):
 * no table in memory, a divergent table read costs more than these few operations */
__device__ __forceinline__ void il_len_sym(uint32_t sy, uint32_t &base, uint32_t &extra)
{
	extra = sy < 8 ? 0u : sy == 28 ? 0u : (sy - 4) >> 2;
	base = sy < 8 ? 3u + sy : sy == 28 ? 258u : ((4u + (sy & 3u)) << extra) + 3u;
}
__device__ __forceinline__ void il_dist_sym(uint32_t ds, uint32_t &base, uint32_t &extra)
{
	extra = ds < 4 ? 0u : (ds >> 1) - 1u;
	base = ds < 4 ? ds + 1u : ((2u + (ds & 1u)) << extra) + 1u;
}

__device__ __constant__ uint8_t il_clc_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

/* the symbol just taken needed bits the member does not have */
/* (bits can only run out once the byte cursor has passed the end of the member: one compare
 * in the common case) */
#define IL_CHECK_TRUNC()  do { if (B.ip > B.iend && lb_avail(B) < 0) { status = LA_ST_GZ_TRUNCATED; goto done; } } while (0)

/*
 * EMIT = false: the member is decoded in place (literals and match copies go straight to the
 * slab).  `only`, if given, restricts the launch to the members it flags.
 *
 * EMIT = true: entropy decoding only.  The lane writes the member's LITERALS densely into a
 * buffer of their own and every match as an 8-byte sequence {literal source, literal run
 * length, output position, distance} -- the very table the LZ4 parse kernel produces -- and
 * lz4_expand_fast_kernel then builds the output in its LDS window (literals with coalesced
 * loads, matches with LDS copies) instead of this lane chasing its own output through global
 * memory one match at a time.  Members the LDS-window kernel cannot take (slot above 64 KiB,
 * more than LA_INFLATE_MAXSEQ matches) are flagged in E.todo for an EMIT = false launch.
 */
template <bool EMIT>
__global__ __launch_bounds__(IL_THREADS) void inflate_lanes_kernel(const uint8_t *__restrict__ src,
    uint64_t src_bytes, const la_gz_member *__restrict__ members, uint32_t n, uint8_t *dst,
    uint64_t dst_cap, la_gz_result *__restrict__ results, uint8_t *scratch, const uint32_t *__restrict__ only,
    la_inflate_emit E)
{
	__shared__ il_lds T;
	const int tid = threadIdx.x;
	const uint32_t mi = blockIdx.x * IL_THREADS + tid;
	if (mi >= n)
		return;
	if (!EMIT && only && only[mi] == 0)
		return;
	const la_gz_member m = members[mi];
	if (EMIT) {
		E.dst_off[mi] = m.dst_off;
		E.table_off[mi] = (uint64_t)mi * LA_INFLATE_MAXSEQ;
		if (mi + 1 == n) {
			E.dst_off[n] = m.dst_off + m.dst_cap;
			E.table_off[n] = (uint64_t)n * LA_INFLATE_MAXSEQ;
		}
		if (m.dst_cap > 65536u) {	/* the LDS window holds 64 KiB */
			E.todo[mi] = 1;
			E.xstatus[mi] = 1;
			E.out_len[mi] = 0;
			E.nseq[mi] = 0;
			return;
		}
	}
	/* EMIT state: literal and sequence counts, current literal run */
	uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;	/* literal accumulator */
	uint64_t pend = 0;
	uint32_t pl0 = 0, pl1 = 0, pl2 = 0, pl3 = 0, pl_at = 0, pt_at = 0;	/* stores waiting for a flush point */
	uint64_t pt_a = 0, pt_b = 0;
	bool pl_on = false, pt_on = false;
	uint32_t nl = 0, ns = 0, run_src = 0, run_dst = 0;
	bool overflow = false;
	uint8_t *const litb = EMIT ? E.lit + (uint64_t)mi * 65536u : nullptr;
	uint64_t *const tabp = EMIT ? (uint64_t *)(E.table + (uint64_t)mi * LA_INFLATE_MAXSEQ) : nullptr;
/* Literals leave through a 16-byte register accumulator and table entries in pairs: fewer,
 * wider stores.  (Measured: plain 1-byte / 8-byte stores are 25 % slower -- the memory
 * counter the bit-buffer refills wait on counts stores too.) */
#define IL_PUT_LIT(byte_)                                                                         \
	do {                                                                                      \
		/* the 16-byte accumulator is a shift register: the new byte enters at the top,   \
		 * after 16 of them the first one has reached byte 0 (four funnel shifts, no      \
		 * position-dependent branch) */                                                  \
		q0 = __builtin_amdgcn_alignbit(q1, q0, 8);                                        \
		q1 = __builtin_amdgcn_alignbit(q2, q1, 8);                                        \
		q2 = __builtin_amdgcn_alignbit(q3, q2, 8);                                        \
		q3 = __builtin_amdgcn_alignbit((uint32_t)(byte_), q3, 8);                         \
		nl++;                                                                             \
		if ((nl & 15u) == 0) {                                                            \
			pl0 = q0; pl1 = q1; pl2 = q2; pl3 = q3; pl_at = nl - 16; pl_on = true;        \
		}                                                                                 \
	} while (0)
/* Stores do not leave where they arise: a finished group of sixteen literals and a finished pair of
 * table entries wait in registers for the next FLUSH POINT, right behind a bit-buffer refill.  The
 * refill waits for the prefetch of the refill before it, and the memory counter it waits on counts
 * stores too (in order): a store issued just in front of that wait would be waited for in full,
 * one issued just behind it has a whole literal burst to complete in. */
#define IL_FLUSH()                                                                                \
	do {                                                                                      \
		if (pl_on) { *(uint4 *)(litb + pl_at) = make_uint4(pl0, pl1, pl2, pl3); pl_on = false; } \
		if (pt_on) {                                                                      \
			*(uint4 *)(tabp + pt_at) = make_uint4((uint32_t)pt_a, (uint32_t)(pt_a >> 32), (uint32_t)pt_b, (uint32_t)(pt_b >> 32)); \
			pt_on = false;                                                            \
		}                                                                                 \
	} while (0)
#define IL_PUT_SEQ(off_)                                                                          \
	do {                                                                                      \
		if (ns >= LA_INFLATE_MAXSEQ) { overflow = true; goto done; }                     \
		const uint64_t e_ = (uint64_t)(run_src | ((nl - run_src) << 16)) |                 \
		    ((uint64_t)(run_dst | ((uint32_t)(off_) << 16)) << 32);                       \
		if (ns & 1u) {                                                                    \
			pt_a = pend; pt_b = e_; pt_at = ns - 1; pt_on = true;                         \
		} else                                                                            \
			pend = e_;                                                                \
		ns++;                                                                             \
	} while (0)
	uint8_t *lens = scratch + (uint64_t)mi * IL_SCRATCH_PER_LANE;		/* [320] */
	uint16_t *sorted_ll = (uint16_t *)(lens + 320);				/* [288] */
	uint16_t *sorted_d = sorted_ll + 288;					/* [32] */
	uint32_t status = LA_ST_OK;
	lane_bits B;
	B.s = src + m.src_off;
	B.safe = (const uint8_t *)members;
	B.iend = m.src_len;
	{
		uint64_t room = m.src_off < src_bytes ? src_bytes - m.src_off : 0;
		B.room = room > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)room;
		if (B.iend > B.room) B.iend = B.room;
	}
	lb_start(B);
	uint8_t *d = dst + m.dst_off;
	uint32_t cap = m.dst_cap;
	if (m.dst_off + m.dst_cap > dst_cap)
		cap = m.dst_off < dst_cap ? (uint32_t)(dst_cap - m.dst_off) : 0;
	uint32_t op = 0;
	lane_code CL, CD;
	CL.maxlen = CD.maxlen = 0;
	p16_zero(CL.count); p16_zero(CD.count);

	[[maybe_unused]] const unsigned long long il_t_start = IL_NOW();
	for (;;) {
		[[maybe_unused]] const unsigned long long il_t_hdr = IL_NOW();
		lb_refill(B);
		if (EMIT) IL_FLUSH();
		const uint32_t last = lb_peek(B, 1);
		const uint32_t type = (lb_peek(B, 3) >> 1);
		lb_drop(B, 3);
		IL_CHECK_TRUNC();
		if (type == 0) {
			/* stored: to the byte boundary, LEN / NLEN, raw bytes */
			lb_drop(B, B.bits & 7);
			lb_refill(B);
			const uint32_t v = (uint32_t)B.hold;
			lb_drop(B, 32);
			IL_CHECK_TRUNC();
			const uint32_t len = v & 0xFFFFu, nlen = v >> 16;
			if (len != (nlen ^ 0xFFFFu)) { status = LA_ST_GZ_DATA; goto done; }
			/* whole bytes still in the bit buffer go back to the byte stream */
			B.ip -= B.bits >> 3;
			const uint32_t avail = B.ip < B.iend ? B.iend - B.ip : 0;
			const uint32_t take = len < avail ? len : avail;
			if (op + take > cap) { status = LA_ST_GZ_OUT_FULL; goto done; }
			if (EMIT) {
				for (uint32_t j = 0; j < take; j++) {
					const uint32_t by = B.s[B.ip + j];
					IL_PUT_LIT(by);
					IL_FLUSH();
				}
			} else {
				for (uint32_t j = 0; j < take; j++)
					d[op + j] = B.s[B.ip + j];
			}
			op += take;
			lb_seek(B, B.ip + take);
			if (take < len) { status = LA_ST_GZ_TRUNCATED; goto done; }
		} else if (type == 1 || type == 2) {
			int nlen = 288, ndist = 32;
			if (type == 1) {
				for (int i = 0; i < 320; i++)
					lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
			} else {
				lb_refill(B);
				nlen = (int)lb_peek(B, 5) + 257; lb_drop(B, 5);
				ndist = (int)lb_peek(B, 5) + 1; lb_drop(B, 5);
				const int ncode = (int)lb_peek(B, 4) + 4; lb_drop(B, 4);
				IL_CHECK_TRUNC();
				if (nlen > 286 || ndist > 30) { status = LA_ST_GZ_DATA; goto done; }
				for (int i = 0; i < 19; i++)
					lens[i] = 0;
				for (int i = 0; i < ncode; i++) {
					if (B.bits < 3) lb_refill(B);
					lens[il_clc_order[i]] = (uint8_t)lb_peek(B, 3);
					lb_drop(B, 3);
					IL_CHECK_TRUNC();
				}
				/* code-length code: 19 symbols, <= 7 bits; its fast table borrows the distance table */
				lane_code CC;
				const int e = il_build<DT_BITS, false>(lens, 19, CC, sorted_d, T, tid);
				if (e != 0 && CC.maxlen != 0) { status = LA_ST_GZ_DATA; goto done; }
				int idx = 0;
				uint32_t prev = 0;
				if (CC.maxlen == 0) {
					/* zlib 1.2.11: an all-zero code-length code yields one-bit "length 0" symbols */
					lb_refill(B);
					if (lb_avail(B) < 1) { status = LA_ST_GZ_TRUNCATED; goto done; }
					for (; idx < nlen + ndist; idx++) {
						if (B.bits < 1) lb_refill(B);
						lb_drop(B, 1);
						IL_CHECK_TRUNC();
						lens[idx] = 0;
					}
				} else {
					while (idx < nlen + ndist) {
						lb_refill(B);
						uint32_t used;
						const int sym = il_decode<DT_BITS, false>(B, CC, sorted_d, T, tid, &used);
						IL_CHECK_TRUNC();
						if (sym < 0) { status = LA_ST_GZ_DATA; goto done; }
						if (sym < 16) {
							lens[idx++] = (uint8_t)sym;
							prev = (uint32_t)sym;
							continue;
						}
						int rep;
						uint32_t val = 0;
						if (sym == 16) {
							rep = 3 + (int)lb_peek(B, 2); lb_drop(B, 2);
							IL_CHECK_TRUNC();
							if (idx == 0) { status = LA_ST_GZ_DATA; goto done; }
							val = prev;
						} else if (sym == 17) {
							rep = 3 + (int)lb_peek(B, 3); lb_drop(B, 3);
							IL_CHECK_TRUNC();
						} else {
							rep = 11 + (int)lb_peek(B, 7); lb_drop(B, 7);
							IL_CHECK_TRUNC();
						}
						if (idx + rep > nlen + ndist) { status = LA_ST_GZ_DATA; goto done; }
						for (int t = 0; t < rep; t++)
							lens[idx + t] = (uint8_t)val;
						prev = val;
						idx += rep;
					}
				}
				if (lens[256] == 0) { status = LA_ST_GZ_DATA; goto done; }
			}
			{
				int e = il_build<LL_BITS, true>(lens, nlen, CL, sorted_ll, T, tid);
				if (e < 0 || (e > 0 && CL.maxlen != 1)) { status = LA_ST_GZ_DATA; goto done; }
				e = il_build<DT_BITS, false>(lens + nlen, ndist, CD, sorted_d, T, tid);
				if (e < 0 || (e > 0 && CD.maxlen > 1)) { status = LA_ST_GZ_DATA; goto done; }
			}
			uint32_t whas = 0;
#if IL_CANON
			/* the lanes that run the symbol loop together keep their codes for all of it: which code
			 * lengths exist among them is one wave-uniform word for the whole loop */
#pragma unroll
			for (int k = 1; k <= 15; k++)
				whas |= __ballot((CL.has >> k) & 1u) ? 1u << k : 0u;
#endif
			IL_CNT(2, IL_NOW() - il_t_hdr);
			IL_CNT(3, 1);
			/* ---- symbols ---- */
			for (;;) {
				IL_CNT(6, 1);
				/* 48 bits cover one literal/length code, its extra bits, a distance code and
				 * its extra bits (15 + 5 + 15 + 13): reload only when fewer are left, i.e.
				 * every few symbols instead of every symbol */
				uint32_t used;
				int sym;
				/* Literal burst: up to IL_LIT_BURST literals in a short loop of their own before
				 * the (three times longer) match path is run for the lanes that have reached a
				 * length code.  Most symbols are literals, so without it every lane pays the
				 * match path per literal; unbounded, every lane would wait for the longest
				 * literal run in the wave (measured: 2.5x slower). */
				sym = 0;
				/* ONE refill for the whole burst, taken by every lane (>= 56 bits afterwards):
				 * three literal/length codes are at most 45 bits, so the burst itself needs no
				 * per-symbol refill test -- with 64 lanes that test fires for some lane in
				 * every iteration and the whole wave pays for the refill code each time */
#pragma unroll	/* IL_LIT_ROUNDS bursts back to back before the match path: it is paid per outer iteration */
				for (int rnd = 0; rnd < IL_LIT_ROUNDS; rnd++) {
					lb_refill(B);
					if (EMIT) IL_FLUSH();
#pragma unroll	/* (no loop around the burst: a loop head makes the compiler wait for the prefetch there) */
					for (int burst = 0; burst < IL_LIT_BURST; burst++) {
						sym = il_decode<LL_BITS, true>(B, CL, sorted_ll, T, tid, &used, whas);
						IL_CHECK_TRUNC();
						if (sym < 0) { status = LA_ST_GZ_DATA; goto done; }
						if (sym >= 256)
							break;
						if (op >= cap) { status = LA_ST_GZ_OUT_FULL; goto done; }
						if (EMIT) {
							IL_PUT_LIT((uint32_t)sym);
							op++;
						} else {
							d[op++] = (uint8_t)sym;
						}
					}
					if (sym >= 256)
						break;
				}
				if (sym < 256)
					continue;	/* the burst ended on a literal: next burst */
				if (sym == 256)
					break;
				/* (after a literal run the buffer may hold fewer than the 33 bits the rest needs) */
				lb_refill(B);	/* (unconditional: a branch around it makes the refill wait for its own prefetch, see lb_refill) */
				IL_CNT(7, 1);
				sym -= 257;
				if (sym >= 29) { status = LA_ST_GZ_DATA; goto done; }
				uint32_t xb, bs;
				il_len_sym((uint32_t)sym, bs, xb);
				const uint32_t length = bs + lb_peek(B, xb);
				lb_drop(B, xb);
				IL_CHECK_TRUNC();
				const int ds = il_decode<DT_BITS, false>(B, CD, sorted_d, T, tid, &used);
				IL_CHECK_TRUNC();
				if (ds < 0 || ds >= 30) { status = LA_ST_GZ_DATA; goto done; }
				il_dist_sym((uint32_t)ds, bs, xb);
				const uint32_t dist = bs + lb_peek(B, xb);
				lb_drop(B, xb);
				IL_CHECK_TRUNC();
				if (dist > op) { status = LA_ST_GZ_DATA; goto done; }
				if (op + length > cap) { status = LA_ST_GZ_OUT_FULL; goto done; }
				if (EMIT) {
					IL_PUT_SEQ(dist);
					op += length;
					run_src = nl;
					run_dst = op;
					continue;
				}
				uint8_t *o = d + op;
				const uint8_t *f = o - dist;
				if (dist >= 16 && op + length + 16 <= cap) {
					/* wild 16-byte copies (two loads in flight, then two stores): bytes past
					 * the match are rewritten by what follows */
					for (uint32_t i = 0; i < length; i += 16) {
						uint64_t v0, v1;
						__builtin_memcpy(&v0, f + i, 8);
						__builtin_memcpy(&v1, f + i + 8, 8);
						__builtin_memcpy(o + i, &v0, 8);
						__builtin_memcpy(o + i + 8, &v1, 8);
					}
				} else if (dist >= 8 && op + length + 8 <= cap) {
					for (uint32_t i = 0; i < length; i += 8) {
						uint64_t v;
						__builtin_memcpy(&v, f + i, 8);
						__builtin_memcpy(o + i, &v, 8);
					}
				} else {
					for (uint32_t i = 0; i < length; i++)
						o[i] = f[i];
				}
				op += length;
			}
		} else {
			status = LA_ST_GZ_DATA;
			goto done;
		}
		if (last)
			break;
	}
done:
	if ((tid & 63) == 0) { IL_CNT(0, IL_NOW() - il_t_start); IL_CNT(1, 1); }
	IL_CNT(10, ns); IL_CNT(11, nl);
	if (EMIT) IL_FLUSH();
	{
		/* bytes consumed = up to and including the byte that holds the last bit used */
		uint32_t consumed = B.ip - (B.bits >> 3);
		if (consumed > B.iend) consumed = B.iend;
		la_gz_result r;
		r.status = status;
		r.out_len = op;
		r.consumed = consumed;
		r.crc32 = 0;
		if (EMIT) {
			if (nl - run_src > 0xFFFFu)
				overflow = true;	/* one literal run of 65536 bytes does not fit the 16-bit table field */
			if (overflow) {
				/* more matches than the LDS-window kernel holds: the in-place kernel redoes it */
				E.todo[mi] = 1;
				E.xstatus[mi] = 1;
				E.out_len[mi] = 0;
				E.nseq[mi] = 0;
				return;
			}
			/* whatever was decoded before an error is output too (bytes before the error are
			 * delivered): close the table with the literals after the last match */
			if (nl > run_src && ns < LA_INFLATE_MAXSEQ) {
				const uint64_t e_ = (uint64_t)(run_src | ((nl - run_src) << 16)) | ((uint64_t)run_dst << 32);
				if (ns & 1u)
					*(uint4 *)(tabp + ns - 1) = make_uint4((uint32_t)pend, (uint32_t)(pend >> 32), (uint32_t)e_, (uint32_t)(e_ >> 32));
				else
					pend = e_;
				ns++;
			} else if (nl > run_src) {
				E.todo[mi] = 1;
				E.xstatus[mi] = 1;
				E.out_len[mi] = 0;
				E.nseq[mi] = 0;
				return;
			}
			if (ns & 1u)
				tabp[ns - 1] = pend;
			if (nl & 15u) {
				/* the last, partial group sits at the top of the shift register: bring its
				 * first byte down to byte 0 */
				for (uint32_t k = nl & 15u; k < 16u; k++) {
					q0 = __builtin_amdgcn_alignbit(q1, q0, 8);
					q1 = __builtin_amdgcn_alignbit(q2, q1, 8);
					q2 = __builtin_amdgcn_alignbit(q3, q2, 8);
					q3 >>= 8;
				}
				*(uint4 *)(litb + (nl & ~15u)) = make_uint4(q0, q1, q2, q3);
			}
			la_lz4_block bk;
			bk.src_off = (uint64_t)mi * 65536u;
			bk.src_len = nl;
			bk.dst_cap = cap;
			bk.flags = 0;
			bk.block_sum = 0;
			E.blocks[mi] = bk;
			E.out_len[mi] = op;
			E.nseq[mi] = ns;
			E.xstatus[mi] = LA_ST_OK;
			E.todo[mi] = 0;
		}
		results[mi] = r;
	}
#undef IL_PUT_LIT
#undef IL_PUT_SEQ
#undef IL_FLUSH
}

#ifdef LA_DIAG
extern "C" int la_diag_set_il_counters(void *d_buf)
{
	unsigned long long *p = (unsigned long long *)d_buf;
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(la_diag_il), &p, sizeof(p));
}
#endif

uint64_t la_inflate_lanes_scratch_bytes(uint32_t n)
{
	return (uint64_t)n * IL_SCRATCH_PER_LANE + 256;
}

void la_launch_inflate_lanes(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, uint8_t *d_dst, uint64_t dst_cap, la_gz_result *d_results,
    void *d_scratch, const uint32_t *d_only)
{
	if (n == 0) return;
	la_inflate_emit none = {};
	hipLaunchKernelGGL(inflate_lanes_kernel<false>, dim3((n + IL_THREADS - 1) / IL_THREADS), dim3(IL_THREADS), 0, s,
	    d_src, src_bytes, d_members, n, d_dst, dst_cap, d_results, (uint8_t *)d_scratch, d_only, none);
}

void la_launch_inflate_symbols(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, uint64_t dst_cap, la_gz_result *d_results,
    void *d_scratch, la_inflate_emit E)
{
	if (n == 0) return;
	hipLaunchKernelGGL(inflate_lanes_kernel<true>, dim3((n + IL_THREADS - 1) / IL_THREADS), dim3(IL_THREADS), 0, s,
	    d_src, src_bytes, d_members, n, (uint8_t *)nullptr, dst_cap, d_results, (uint8_t *)d_scratch,
	    (const uint32_t *)nullptr, E);
}
