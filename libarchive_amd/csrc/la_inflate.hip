/*
 * la_inflate.hip -- raw DEFLATE decode for batches of independent gzip members
 * (gfx950).  Replaces the zlib inflate() loop of gzip_filter_read
 * (libarchive/archive_read_support_filter_gzip.c:431-511, inflateInit2(-15)
 * at :363) for a whole table of members per launch, and adds the trailer
 * CRC32/ISIZE check the reference leaves as a TODO (gzip.c:423).
 *
 * One WAVE per member.  A deflate stream is a serial bit chain, so the wave
 * keeps the chain wave-uniform (bit buffer and positions in SGPRs, fed from a
 * 512-byte register window of the compressed bytes via v_readlane) and uses
 * its lanes where the format allows it:
 *   - Huffman tables live in LDS (per wave: 10-bit lit/len and 8-bit distance
 *     lookup tables plus the canonical count/symbol arrays for longer codes);
 *   - literals are gathered in a 64-entry lane buffer and stored 64 at a time;
 *   - match copies are wave-wide (64 bytes per step) with the same
 *     store->load visibility rule as the general LZ4 kernel.
 * Accept/reject rules follow zlib 1.2.11 (see oracle/orc_inflate.c).
 * Byte/integer work; no MFMA.
 *
 * Speed (one 64 KiB member, ms): 21.8 at first -- the bit reader and the window lived in SCRATCH memory
 * because inflate_codes() was an out-of-line function taking them by reference (scratch_load /
 * scratch_store + s_waitcnt vmcnt(0) on every refill); inlined: 9.1.  Length / distance bases by
 * arithmetic instead of constant-table loads and no division for non-overlapping copies: see
 * DESIGN.md §5.  What is left (ISA of this build): the chain state (bit buffer, bit count, byte cursor,
 * the decoded symbol) still sits in VGPRs and every uniform `if` is an exec-mask sequence (about 900
 * s_and_saveexec in the kernel against a dozen s_cbranch_scc): the compiler's uniformity analysis loses
 * the state somewhere; readfirstlane on the member record, on the slow-path LDS loads and on maxlen did
 * not bring it back, and forcing it (readfirstlane on the whole chain state once per symbol: 43 scalar
 * branches instead of 7) made the kernel 7 % slower -- so the exec-mask form is not what the ~1 800
 * cycles per symbol are made of either, and neither are the table builds (huff_build inlined, no scratch
 * left: 9.08 against 9.16 ms).  Cycle counters in the diagnostic build (tools/exp_inflate_stamps.py, zlib -6
 * members of the bench stream): 24 216 symbols per 64 KiB member of which 2 251 matches, 930 cycles per
 * symbol; literal/length decode with refill 357 per symbol (38 %), literal bookkeeping 244 per symbol (26 %),
 * length + distance decode 1 074 per match (11 %), flush + copy 648 per match (6.5 %), block headers and
 * table builds 18 %.  Nine symbols in ten are literals at ~600 cycles each: two literals per table look
 * (pair table) and a chain that really lives in SGPRs are what to build next.  Tried and not kept: the output in a 64 KiB LDS ring per wave (match copies LDS to
 * LDS, 16-byte drains to the slab) instead of store / fence / load through global memory: 9.8 ms for one
 * member and, with only two waves per CU, 77.6 instead of 12.5 ms for 4 096 members.
 */
#include "la_dev.h"

/* Diagnostic build only (make diag, -DLA_DIAG): per-member cycle totals of the symbol loop's parts go to a
 * buffer of their own (8 x u64 per member); no output value depends on them. */
#ifdef LA_DIAG
__device__ unsigned long long *la_inf_diag;
extern "C" int la_diag_set_inflate_stamps(void *d_buf)
{
	unsigned long long *p = (unsigned long long *)d_buf;
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(la_inf_diag), &p, sizeof(p));
}
struct inf_diag { unsigned long long t, acc[8]; };
#define DIAG_DECL      inf_diag DG = {}
#define DIAG_ARG       , inf_diag &DG
#define DIAG_PASS      , DG
#define DIAG_T0()      (DG.t = __builtin_readcyclecounter())
#define DIAG_ACC(k)    do { unsigned long long n_ = __builtin_readcyclecounter(); DG.acc[k] += n_ - DG.t; DG.t = n_; } while (0)
#define DIAG_CNT(k)    (DG.acc[k] += 1)
#else
#define DIAG_DECL      do { } while (0)
#define DIAG_ARG
#define DIAG_PASS
#define DIAG_T0()      do { } while (0)
#define DIAG_ACC(k)    do { } while (0)
#define DIAG_CNT(k)    do { } while (0)
#endif

#define INF_WAVES_PER_WG 4
#define LL_FAST_BITS 10
#define D_FAST_BITS  8

struct inf_tables {	/* one per wave, in LDS */
	uint16_t ll_fast[1 << LL_FAST_BITS];	/* (symbol << 4) | length, 0 = long code / unassigned */
	uint16_t d_fast[1 << D_FAST_BITS];
	uint16_t ll_count[16], d_count[16];
	uint16_t ll_symbol[288], d_symbol[32];
	uint8_t  lens[320];
	uint16_t ll_maxlen, d_maxlen;
};

struct bitreader {
	uint64_t hold;
	int bits;
	const uint8_t *ip, *iend;
	src_window W;
};

__device__ __forceinline__ void br_refill(bitreader &B, int lane)
{
	if (B.bits <= 32) {
		int nbytes = (int)(B.iend - B.ip);
		if (nbytes > 4) nbytes = 4;
		if (nbytes > 0) {
			uint32_t v = win_u32(B.W, B.ip, lane);
			if (nbytes < 4)
				v &= (1u << (8 * nbytes)) - 1u;
			B.hold |= (uint64_t)v << B.bits;
			B.bits += 8 * nbytes;
			B.ip += nbytes;
		}
	}
}
__device__ __forceinline__ bool br_need(bitreader &B, int n, int lane)
{
	br_refill(B, lane);
	return B.bits >= n;
}
__device__ __forceinline__ uint32_t br_take(bitreader &B, int n)
{
	uint32_t v = (uint32_t)(B.hold & ((1ull << n) - 1ull));
	B.hold >>= n;
	B.bits -= n;
	return v;
}

__device__ __constant__ uint8_t c_clc_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

__device__ __forceinline__ uint32_t bitrev(uint32_t v, int n) { return __builtin_bitreverse32(v) >> (32 - n); }

/*
 * Canonical Huffman table from lens[0..n): counts, sorted symbols and the fast
 * lookup table.  All lanes run it redundantly on uniform values; lane 0 stores.
 * Returns 0 complete, >0 incomplete, <0 over-subscribed.
 */
__device__ int huff_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *symbol,
    uint16_t *fast, int fast_bits, uint16_t *maxlen_out, int lane)
{
	uint32_t cnt[16];
	for (int l = 0; l < 16; l++) cnt[l] = 0;
	/* lane-parallel histogram */
	for (int i = lane; i < n; i += LA_WAVE) {
		int l = lens[i];
		for (int k = 1; k < 16; k++) cnt[k] += (l == k);
	}
	int left = 1, maxlen = 0;
	uint32_t offs[16], code[16];
	offs[0] = 0; offs[1] = 0; code[0] = 0;
	uint32_t c = 0;
	for (int l = 1; l < 16; l++) {
		uint32_t t = cnt[l];
		for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
		cnt[l] = t;
		if (t) maxlen = l;
		left = left * 2 - (int)t;
		if (left < 0) left = -100000;	/* stays negative */
	}
	for (int l = 1; l < 16; l++) {
		c = (c + (l > 1 ? cnt[l - 1] : 0)) << 1;
		code[l] = c;
		if (l < 15) offs[l + 1] = offs[l] + cnt[l];
	}
	if (lane == 0) {
		for (int l = 0; l < 16; l++) count[l] = (uint16_t)cnt[l];
		*maxlen_out = (uint16_t)maxlen;
	}
	for (int i = lane; i < (1 << fast_bits); i += LA_WAVE)
		fast[i] = 0;
	if (left < 0)
		return -1;
	/* sorted symbol list and fast table: serial in symbol order (stable), lane 0 */
	if (lane == 0) {
		uint32_t next_off[16], next_code[16];
		for (int l = 0; l < 16; l++) { next_off[l] = offs[l]; next_code[l] = code[l]; }
		for (int s = 0; s < n; s++) {
			int l = lens[s];
			if (l == 0) continue;
			symbol[next_off[l]++] = (uint16_t)s;
			uint32_t cw = next_code[l]++;
			if (l <= fast_bits) {
				uint32_t r = bitrev(cw, l);
				uint16_t e = (uint16_t)((s << 4) | l);
				for (uint32_t idx = r; idx < (1u << fast_bits); idx += (1u << l))
					fast[idx] = e;
			}
		}
	}
	return left;
}

/* decode one symbol: >= 0 symbol, -1 input exhausted, -2 unassigned code */
__device__ __forceinline__ int huff_decode(bitreader &B, const uint16_t *fast, int fast_bits,
    const uint16_t *count, const uint16_t *symbol, int maxlen, int lane)
{
	br_refill(B, lane);
	uint32_t e = fast[(uint32_t)B.hold & ((1u << fast_bits) - 1u)];
	e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
	int l = (int)(e & 15);
	if (l != 0) {
		if (l > B.bits)
			return -1;
		B.hold >>= l;
		B.bits -= l;
		return (int)(e >> 4);
	}
	/* long or unassigned code: canonical walk, one bit at a time */
	int codev = 0, first = 0, index = 0;
	int ml = maxlen ? maxlen : 1;
	for (int k = 1; k <= ml; k++) {
		if (B.bits < 1) {
			br_refill(B, lane);
			if (B.bits < 1)
				return -1;
		}
		codev |= (int)(B.hold & 1);
		B.hold >>= 1;
		B.bits -= 1;
		int cn = count[k];
		if (codev - cn < first)
			return symbol[index + (codev - first)];
		index += cn;
		first += cn;
		first <<= 1;
		codev <<= 1;
	}
	return -2;
}

struct out_state {
	uint8_t *d;
	uint32_t op, cap;
	uint32_t visible;	/* bytes [0, visible) are known visible to this wave's loads */
	uint32_t npend;		/* literals waiting in the lane buffer */
	uint32_t litbuf;	/* lane i holds pending literal i */
};

__device__ __forceinline__ void out_flush(out_state &O, int lane)
{
	if (O.npend) {
		if ((uint32_t)lane < O.npend)
			O.d[O.op + lane] = (uint8_t)O.litbuf;
		O.op += O.npend;
		O.npend = 0;
	}
}

/* returns LA_ST_OK, LA_ST_GZ_DATA, LA_ST_GZ_TRUNCATED or LA_ST_GZ_OUT_FULL */
__device__ __forceinline__ uint32_t inflate_codes(bitreader &B, out_state &O, const inf_tables *T, bool fixed,
    const uint16_t *fx_ll_fast, int lane DIAG_ARG)
{
	(void)fixed; (void)fx_ll_fast;
	for (;;) {
		DIAG_T0();
		int sym = huff_decode(B, T->ll_fast, LL_FAST_BITS, T->ll_count, T->ll_symbol, T->ll_maxlen, lane);
		DIAG_ACC(0); DIAG_CNT(6);
		if (sym == -1) return LA_ST_GZ_TRUNCATED;
		if (sym < 0) return LA_ST_GZ_DATA;
		if (sym < 256) {
			if (O.op + O.npend >= O.cap) return LA_ST_GZ_OUT_FULL;
			if ((uint32_t)lane == O.npend)
				O.litbuf = (uint32_t)sym;
			O.npend++;
			if (O.npend == LA_WAVE)
				out_flush(O, lane);
			DIAG_ACC(1);
			continue;
		}
		if (sym == 256)
			return LA_ST_OK;
		sym -= 257;
		if (sym >= 29) return LA_ST_GZ_DATA;
		/* base value and extra bits by arithmetic (RFC 1951 3.2.5), as in la_inflate_lanes.hip: a
		 * constant-table load is a scalar memory round trip in the middle of the serial chain */
		int xb = sym < 8 ? 0 : sym == 28 ? 0 : (sym - 4) >> 2;
		if (!br_need(B, xb, lane)) return LA_ST_GZ_TRUNCATED;
		uint32_t length = (sym < 8 ? 3u + (uint32_t)sym : sym == 28 ? 258u : ((4u + ((uint32_t)sym & 3u)) << xb) + 3u) + br_take(B, xb);
		int ds = huff_decode(B, T->d_fast, D_FAST_BITS, T->d_count, T->d_symbol, T->d_maxlen, lane);
		if (ds == -1) return LA_ST_GZ_TRUNCATED;
		if (ds < 0 || ds >= 30) return LA_ST_GZ_DATA;
		xb = ds < 4 ? 0 : (ds >> 1) - 1;
		if (!br_need(B, xb, lane)) return LA_ST_GZ_TRUNCATED;
		uint32_t dist = (ds < 4 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << xb) + 1u) + br_take(B, xb);
		DIAG_ACC(2); DIAG_CNT(7);
		out_flush(O, lane);
		if (dist > O.op) return LA_ST_GZ_DATA;
		if (O.op + length > O.cap) return LA_ST_GZ_OUT_FULL;
		/* wave-wide copy; the modulo form only reads bytes below op */
		uint32_t span = length < dist ? length : dist;
		if (O.op - dist + span > O.visible) {
			wave_mem_fence();
			O.visible = O.op;
		}
		if (dist >= length) {	/* (wave-uniform) no overlap: no division */
			for (uint32_t j = (uint32_t)lane; j < length; j += LA_WAVE)
				O.d[O.op + j] = O.d[O.op - dist + j];
		} else {
			for (uint32_t j = (uint32_t)lane; j < length; j += LA_WAVE)
				O.d[O.op + j] = O.d[O.op - dist + j % dist];
		}
		O.op += length;
		DIAG_ACC(3);
	}
}

__global__ __launch_bounds__(64 * INF_WAVES_PER_WG) void inflate_kernel(const uint8_t *__restrict__ src,
    uint64_t src_bytes, const la_gz_member *__restrict__ members, uint32_t n, uint8_t *dst,
    uint64_t dst_cap, la_gz_result *__restrict__ results)
{
	__shared__ inf_tables tabs[INF_WAVES_PER_WG];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const uint32_t mi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * INF_WAVES_PER_WG + wv));
	if (mi >= n)
		return;
	inf_tables *T = &tabs[wv];
	const la_gz_member m = members[mi];
	uint32_t status = LA_ST_OK;
	DIAG_DECL;
#ifdef LA_DIAG
	const unsigned long long dg_start = __builtin_readcyclecounter();
#endif
	bitreader B;
	B.hold = 0; B.bits = 0;
	B.ip = src + m.src_off;
	B.iend = B.ip + m.src_len;
	if (m.src_off + m.src_len > src_bytes)
		B.iend = src + src_bytes;
	B.W.limit = src + src_bytes;
	win_reset(B.W, B.ip, lane);
	out_state O;
	O.d = dst + m.dst_off;
	O.op = 0; O.visible = 0; O.npend = 0; O.litbuf = 0;
	O.cap = m.dst_cap;
	if (m.dst_off + m.dst_cap > dst_cap)
		O.cap = m.dst_off < dst_cap ? (uint32_t)(dst_cap - m.dst_off) : 0;

	for (;;) {
		if (!br_need(B, 3, lane)) { status = LA_ST_GZ_TRUNCATED; break; }
		int last = (int)br_take(B, 1);
		int type = (int)br_take(B, 2);
		if (type == 0) {
			br_take(B, B.bits & 7);
			/* LEN / NLEN */
			if (!br_need(B, 32, lane)) {
				/* br_refill only tops up to > 32 bits; try once more for exactly 32 */
				br_refill(B, lane);
				if (B.bits < 32) { status = LA_ST_GZ_TRUNCATED; break; }
			}
			uint32_t v = br_take(B, 32);
			uint32_t len = v & 0xffff, nlen = v >> 16;
			if (len != (nlen ^ 0xffff)) { status = LA_ST_GZ_DATA; break; }
			out_flush(O, lane);
			/* whole bytes still in the bit buffer go back to the byte stream */
			B.ip -= B.bits >> 3;
			B.bits = 0; B.hold = 0;
			uint32_t avail = (uint32_t)(B.iend - B.ip);
			uint32_t take = len < avail ? len : avail;
			if (O.op + take > O.cap) { status = LA_ST_GZ_OUT_FULL; break; }
			for (uint32_t j = (uint32_t)lane; j < take; j += LA_WAVE)
				O.d[O.op + j] = B.ip[j];
			O.op += take;
			B.ip += take;
			if (take < len) { status = LA_ST_GZ_TRUNCATED; break; }
		} else if (type == 1) {
			/* fixed code: lengths 8/9/7/8 for 288 symbols, 5 bits for 32 distance symbols */
			for (int i = lane; i < 320; i += LA_WAVE)
				T->lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
			huff_build(T->lens, 288, T->ll_count, T->ll_symbol, T->ll_fast, LL_FAST_BITS, &T->ll_maxlen, lane);
			huff_build(T->lens + 288, 32, T->d_count, T->d_symbol, T->d_fast, D_FAST_BITS, &T->d_maxlen, lane);
			status = inflate_codes(B, O, T, true, nullptr, lane DIAG_PASS);
			if (status != LA_ST_OK) break;
		} else if (type == 2) {
			if (!br_need(B, 14, lane)) { status = LA_ST_GZ_TRUNCATED; break; }
			int nlen = (int)br_take(B, 5) + 257;
			int ndist = (int)br_take(B, 5) + 1;
			int ncode = (int)br_take(B, 4) + 4;
			if (nlen > 286 || ndist > 30) { status = LA_ST_GZ_DATA; break; }
			for (int i = lane; i < 320; i += LA_WAVE)
				T->lens[i] = 0;
			bool trunc = false;
			for (int i = 0; i < ncode; i++) {
				if (!br_need(B, 3, lane)) { trunc = true; break; }
				uint32_t v = br_take(B, 3);
				if (lane == 0) T->lens[c_clc_order[i]] = (uint8_t)v;
			}
			if (trunc) { status = LA_ST_GZ_TRUNCATED; break; }
			/* the code-length code reuses the distance table slots (19 symbols, 7-bit codes) */
			int e = huff_build(T->lens, 19, T->d_count, T->d_symbol, T->d_fast, D_FAST_BITS, &T->d_maxlen, lane);
			int cl_max = T->d_maxlen;
			cl_max = __builtin_amdgcn_readfirstlane(cl_max);
			if (e != 0 && cl_max != 0) { status = LA_ST_GZ_DATA; break; }
			/* code lengths are decoded into a scratch copy first (lens[] holds the CL code) */
			int idx = 0;
			uint32_t prev = 0;
			uint8_t *L = T->lens;		/* overwritten in place AFTER the CL table is built */
			if (cl_max == 0) {
				/* zlib 1.2.11: an all-zero code-length code yields one-bit "length 0" symbols */
				if (!br_need(B, 1, lane)) { status = LA_ST_GZ_TRUNCATED; break; }
				for (; idx < nlen + ndist; idx++) {
					if (!br_need(B, 1, lane)) break;
					br_take(B, 1);
				}
				if (idx < nlen + ndist) { status = LA_ST_GZ_TRUNCATED; break; }
				for (int i = lane; i < 320; i += LA_WAVE) L[i] = 0;
			} else {
				for (int i = lane; i < 320; i += LA_WAVE) L[i] = 0;
				while (idx < nlen + ndist) {
					int sym = huff_decode(B, T->d_fast, D_FAST_BITS, T->d_count, T->d_symbol, cl_max, lane);
					if (sym == -1) { status = LA_ST_GZ_TRUNCATED; break; }
					if (sym < 0) { status = LA_ST_GZ_DATA; break; }
					if (sym < 16) {
						if (lane == 0) L[idx] = (uint8_t)sym;
						prev = (uint32_t)sym;
						idx++;
						continue;
					}
					int rep;
					uint32_t val = 0;
					if (sym == 16) {
						if (!br_need(B, 2, lane)) { status = LA_ST_GZ_TRUNCATED; break; }
						if (idx == 0) { status = LA_ST_GZ_DATA; break; }
						val = prev;
						rep = 3 + (int)br_take(B, 2);
					} else if (sym == 17) {
						if (!br_need(B, 3, lane)) { status = LA_ST_GZ_TRUNCATED; break; }
						rep = 3 + (int)br_take(B, 3);
					} else {
						if (!br_need(B, 7, lane)) { status = LA_ST_GZ_TRUNCATED; break; }
						rep = 11 + (int)br_take(B, 7);
					}
					if (idx + rep > nlen + ndist) { status = LA_ST_GZ_DATA; break; }
					if (lane < rep) L[idx + lane] = (uint8_t)val;
					if (rep > 64 && lane + 64 < rep) L[idx + lane + 64] = (uint8_t)val;
					if (rep > 128 && lane + 128 < rep) L[idx + lane + 128] = (uint8_t)val;
					prev = val;
					idx += rep;
				}
				if (status != LA_ST_OK) break;
			}
			if (L[256] == 0) { status = LA_ST_GZ_DATA; break; }
			e = huff_build(L, nlen, T->ll_count, T->ll_symbol, T->ll_fast, LL_FAST_BITS, &T->ll_maxlen, lane);
			int llm = __builtin_amdgcn_readfirstlane((int)T->ll_maxlen);
			if (e < 0 || (e > 0 && llm != 1)) { status = LA_ST_GZ_DATA; break; }
			e = huff_build(L + nlen, ndist, T->d_count, T->d_symbol, T->d_fast, D_FAST_BITS, &T->d_maxlen, lane);
			int dm = __builtin_amdgcn_readfirstlane((int)T->d_maxlen);
			if (e < 0 || (e > 0 && dm > 1)) { status = LA_ST_GZ_DATA; break; }
			status = inflate_codes(B, O, T, false, nullptr, lane DIAG_PASS);
			if (status != LA_ST_OK) break;
		} else {
			status = LA_ST_GZ_DATA;
			break;
		}
		if (last)
			break;
	}
	/* what zlib would have emitted before noticing: whole symbols, stored data bytewise */
	if (status != LA_ST_GZ_OUT_FULL)
		out_flush(O, lane);
#ifdef LA_DIAG
	if (lane == 0 && la_inf_diag) {
		for (int k = 0; k < 8; k++)
			la_inf_diag[(size_t)mi * 8 + k] = DG.acc[k];
		la_inf_diag[(size_t)mi * 8 + 4] = __builtin_readcyclecounter() - dg_start;	/* whole member */
	}
#endif
	uint32_t consumed = (uint32_t)(B.ip - (src + m.src_off)) - (uint32_t)(B.bits >> 3);
	if (lane == 0) {
		la_gz_result r;
		r.status = status;
		r.out_len = O.op;
		r.consumed = consumed;
		r.crc32 = 0;
		results[mi] = r;
	}
}

void la_launch_inflate(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, uint8_t *d_dst, uint64_t dst_cap, la_gz_result *d_results)
{
	if (n == 0) return;
	hipLaunchKernelGGL(inflate_kernel, dim3((n + INF_WAVES_PER_WG - 1) / INF_WAVES_PER_WG),
	    dim3(64 * INF_WAVES_PER_WG), 0, s, d_src, src_bytes, d_members, n, d_dst, dst_cap, d_results);
}

