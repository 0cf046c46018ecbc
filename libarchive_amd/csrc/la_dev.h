/*
 * la_dev.h -- shared device-side helpers and the internal launch interface
 * between the kernel translation units and the C-ABI layer (la_api.hip).
 * gfx950 only: wave = 64 lanes, 160 KiB LDS per CU.
 */
#ifndef LA_DEV_H
#define LA_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/la_gpu.h"

#define LA_WAVE 64

/* ---- unaligned little-endian loads (gfx950 global memory is byte addressable
 * and the HSA ABI runs with unaligned access enabled; the compiler emits one
 * global_load_dword[x4] for these) ---- */
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}
__device__ __forceinline__ uint4 ld_u128(const uint8_t *p)
{
	uint4 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r)
{
	return __builtin_rotateleft32(x, r);
}

/* ---- XXH32 (libarchive/xxhash.c:189-193 constants, :245-291 arithmetic) ---- */
#define XXH_P1 0x9E3779B1u
#define XXH_P2 0x85EBCA77u
#define XXH_P3 0xC2B2AE3Du
#define XXH_P4 0x27D4EB2Fu
#define XXH_P5 0x165667B1u

__device__ __forceinline__ uint32_t xxh_round(uint32_t acc, uint32_t lane)
{
	return rotl32(acc + lane * XXH_P2, 13) * XXH_P1;
}
__device__ __forceinline__ uint32_t xxh_avalanche(uint32_t h)
{
	h ^= h >> 15; h *= XXH_P2;
	h ^= h >> 13; h *= XXH_P3;
	h ^= h >> 16;
	return h;
}
/* one whole hash computed by ONE lane (many-hash kernels run one hash per lane).
 * The stripe loop is unrolled four times so that a lane keeps 64 bytes of loads
 * in flight: per-lane streams are latency bound, not issue bound. */
__device__ __forceinline__ uint32_t xxh32_lane(const uint8_t *p, uint32_t len, uint32_t seed)
{
	const uint8_t *end = p + len;
	uint32_t h;
	if (len >= 16) {
		uint32_t v1 = seed + XXH_P1 + XXH_P2, v2 = seed + XXH_P2, v3 = seed, v4 = seed - XXH_P1;
		while (p + 64 <= end) {
			uint4 a = ld_u128(p), b = ld_u128(p + 16), c = ld_u128(p + 32), d = ld_u128(p + 48);
			v1 = xxh_round(v1, a.x); v2 = xxh_round(v2, a.y); v3 = xxh_round(v3, a.z); v4 = xxh_round(v4, a.w);
			v1 = xxh_round(v1, b.x); v2 = xxh_round(v2, b.y); v3 = xxh_round(v3, b.z); v4 = xxh_round(v4, b.w);
			v1 = xxh_round(v1, c.x); v2 = xxh_round(v2, c.y); v3 = xxh_round(v3, c.z); v4 = xxh_round(v4, c.w);
			v1 = xxh_round(v1, d.x); v2 = xxh_round(v2, d.y); v3 = xxh_round(v3, d.z); v4 = xxh_round(v4, d.w);
			p += 64;
		}
		while (p + 16 <= end) {
			uint4 x = ld_u128(p);
			v1 = xxh_round(v1, x.x);
			v2 = xxh_round(v2, x.y);
			v3 = xxh_round(v3, x.z);
			v4 = xxh_round(v4, x.w);
			p += 16;
		}
		h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
	} else {
		h = seed + XXH_P5;
	}
	h += len;
	while (p + 4 <= end) {
		h = rotl32(h + ld_u32(p) * XXH_P3, 17) * XXH_P4;
		p += 4;
	}
	while (p < end) {
		h = rotl32(h + (uint32_t)(*p) * XXH_P5, 11) * XXH_P1;
		p++;
	}
	return xxh_avalanche(h);
}

/* one hash computed by FOUR adjacent lanes (lane j of the quad owns accumulator
 * v(j+1) and reads the j-th dword of every 16-byte stripe, so a quad's load is one
 * contiguous 16 bytes).  Used for the long content-checksum chains, where there
 * are few hashes and each is long.  Result valid in all four lanes. */
__device__ __forceinline__ uint32_t xxh32_quad(const uint8_t *p, uint32_t len, uint32_t seed, int j)
{
	const uint8_t *end = p + len;
	uint32_t h;
	if (len >= 16) {
		uint32_t v = (j == 0) ? seed + XXH_P1 + XXH_P2 : (j == 1) ? seed + XXH_P2 : (j == 2) ? seed : seed - XXH_P1;
		const uint8_t *q = p + 4 * j;
		/* 512 bytes (32 stripes) of loads in flight per quad: the chain itself is short,
		 * the stream must not wait for memory.  (Requesting the next batch before this one is
		 * chained -- software prefetch with a second register set -- made this loop twice as
		 * slow as compiled, 14 vs 7 ms per step; it is kept in the 16-lane form below only.) */
		while (p + 512 <= end) {
			uint32_t x[32];
#pragma unroll
			for (int t = 0; t < 32; t++)
				x[t] = ld_u32(q + 16 * t);
#pragma unroll
			for (int t = 0; t < 32; t++)
				v = xxh_round(v, x[t]);
			p += 512; q += 512;
		}
		while (p + 128 <= end) {
			uint32_t x0 = ld_u32(q), x1 = ld_u32(q + 16), x2 = ld_u32(q + 32), x3 = ld_u32(q + 48);
			uint32_t x4 = ld_u32(q + 64), x5 = ld_u32(q + 80), x6 = ld_u32(q + 96), x7 = ld_u32(q + 112);
			v = xxh_round(v, x0); v = xxh_round(v, x1); v = xxh_round(v, x2); v = xxh_round(v, x3);
			v = xxh_round(v, x4); v = xxh_round(v, x5); v = xxh_round(v, x6); v = xxh_round(v, x7);
			p += 128; q += 128;
		}
		while (p + 16 <= end) {
			v = xxh_round(v, ld_u32(q));
			p += 16; q += 16;
		}
		const int rot = (j == 0) ? 1 : (j == 1) ? 7 : (j == 2) ? 12 : 18;
		uint32_t t = rotl32(v, rot);
		t += __shfl_xor(t, 1, 64);
		t += __shfl_xor(t, 2, 64);
		h = t;
	} else {
		h = seed + XXH_P5;
	}
	h += len;
	while (p + 4 <= end) {
		h = rotl32(h + ld_u32(p) * XXH_P3, 17) * XXH_P4;
		p += 4;
	}
	while (p < end) {
		h = rotl32(h + (uint32_t)(*p) * XXH_P5, 11) * XXH_P1;
		p++;
	}
	return xxh_avalanche(h);
}

/* The same hash by SIXTEEN adjacent lanes (one DPP row).  A single XXH32 accumulator is a serial
 * chain  v = rotl(v + x * P2, 13) * P1  whose two 32-bit multiplies run at quarter rate; in the
 * four-lane form every lane pays both per stripe.  Here lane 4t + j loads dword j of stripe t of
 * each 64-byte group (one coalesced 64-byte load per row), ONE multiply instruction turns all of
 * them into products, and lanes 0..3 (accumulator j) chain over the four stripes pulling the
 * products of the other lanes in with DPP row shifts: one multiply on the chain per stripe
 * instead of two, 29 instead of 62 cycles per stripe.  Only a quarter as many hashes share an
 * instruction, so this form is for hashes nothing else hides (the last slice of a batch, frames
 * that cross batches), not for bulk.  Result valid in lanes 0..3 of the row. */
#define XXH_ROW_BATCH 32
__device__ __forceinline__ uint32_t xxh_chain_step(uint32_t v, uint32_t prod)
{
	return rotl32(v + prod, 13) * XXH_P1;
}
__device__ __forceinline__ uint32_t xxh32_row(const uint8_t *p, uint32_t len, uint32_t seed, int l)
{
	const int j = l & 3;
	const uint8_t *end = p + len;
	uint32_t h;
	if (len >= 16) {
		uint32_t v = (j == 0) ? seed + XXH_P1 + XXH_P2 : (j == 1) ? seed + XXH_P2 : (j == 2) ? seed : seed - XXH_P1;
		const uint8_t *q = p + 4 * l;
		/* XXH_ROW_BATCH groups of four stripes per batch (2 KiB per row); the next batch is
		 * requested before this one is chained, and chaining a batch takes about as long as a
		 * load does */
#define XXH_ROW_CHAIN(xg_)                                                                                        \
		do {                                                                                              \
			const uint32_t pr_ = (xg_) * XXH_P2;                                                      \
			v = xxh_chain_step(v, pr_);                                                               \
			v = xxh_chain_step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pr_, 0x104, 0xf, 0xf, false));	/* row_shl:4 */ \
			v = xxh_chain_step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pr_, 0x108, 0xf, 0xf, false));	/* row_shl:8 */ \
			v = xxh_chain_step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pr_, 0x10c, 0xf, 0xf, false));	/* row_shl:12 */ \
		} while (0)
		if (p + XXH_ROW_BATCH * 64 <= end) {
			uint32_t x[XXH_ROW_BATCH];
#pragma unroll
			for (int g = 0; g < XXH_ROW_BATCH; g++)
				x[g] = ld_u32(q + 64 * g);
			p += XXH_ROW_BATCH * 64; q += XXH_ROW_BATCH * 64;
			while (p + XXH_ROW_BATCH * 64 <= end) {
				uint32_t y[XXH_ROW_BATCH];
#pragma unroll
				for (int g = 0; g < XXH_ROW_BATCH; g++)
					y[g] = ld_u32(q + 64 * g);
#pragma unroll
				for (int g = 0; g < XXH_ROW_BATCH; g++)
					XXH_ROW_CHAIN(x[g]);
#pragma unroll
				for (int g = 0; g < XXH_ROW_BATCH; g++)
					x[g] = y[g];
				p += XXH_ROW_BATCH * 64; q += XXH_ROW_BATCH * 64;
			}
#pragma unroll
			for (int g = 0; g < XXH_ROW_BATCH; g++)
				XXH_ROW_CHAIN(x[g]);
		}
		while (p + 64 <= end) {
			XXH_ROW_CHAIN(ld_u32(q));
			p += 64; q += 64;
		}
#undef XXH_ROW_CHAIN
		while (p + 16 <= end) {	/* fewer than four stripes left: lanes 0..3 on their own */
			v = xxh_round(v, ld_u32(p + 4 * j));
			p += 16;
		}
		const int rot = (j == 0) ? 1 : (j == 1) ? 7 : (j == 2) ? 12 : 18;
		uint32_t t = rotl32(v, rot);
		t += __shfl_xor(t, 1, 64);
		t += __shfl_xor(t, 2, 64);
		h = t;
	} else {
		h = seed + XXH_P5;
	}
	h += len;
	while (p + 4 <= end) {
		h = rotl32(h + ld_u32(p) * XXH_P3, 17) * XXH_P4;
		p += 4;
	}
	while (p < end) {
		h = rotl32(h + (uint32_t)(*p) * XXH_P5, 11) * XXH_P1;
		p++;
	}
	return xxh_avalanche(h);
}

/* 512-byte register window over the compressed payload: lane l holds the
 * aligned dwords at base+4l (w0) and base+256+4l (w1). */
struct src_window {
	const uint8_t *base;	/* 4-byte aligned, wave-uniform */
	const uint8_t *limit;	/* one past the last readable byte of the whole source image */
	uint32_t w0, w1;
};

__device__ __forceinline__ uint32_t win_load(const src_window &W, const uint8_t *p)
{
	/* p is 4-aligned.  A dword that holds at least one byte of the image lies in
	 * the same page as that byte, so loading it whole cannot fault; dwords
	 * entirely past the image are never touched. */
	return (p < W.limit) ? *(const uint32_t *)p : 0u;
}

__device__ __forceinline__ void win_reset(src_window &W, const uint8_t *q, int lane)
{
	W.base = (const uint8_t *)((uintptr_t)q & ~(uintptr_t)3);
	W.w0 = win_load(W, W.base + 4 * lane);
	W.w1 = win_load(W, W.base + 256 + 4 * lane);
}

/* byte at q (wave-uniform address, q >= W.base) */
__device__ __forceinline__ uint32_t win_byte(src_window &W, const uint8_t *q, int lane)
{
	uint32_t d = (uint32_t)(q - W.base);
	if (d >= 256) {
		if (d < 512) {
			W.base += 256;
			W.w0 = W.w1;
			W.w1 = win_load(W, W.base + 256 + 4 * lane);
			d -= 256;
		} else {
			win_reset(W, q, lane);
			d = (uint32_t)(q - W.base);
		}
	}
	uint32_t dw = (uint32_t)__builtin_amdgcn_readlane((int)W.w0, (int)(d >> 2));
	return (dw >> ((d & 3) * 8)) & 0xffu;
}

__device__ __forceinline__ void wave_mem_fence()
{
	/* make this wave's earlier global stores visible to its own later loads
	 * (same CU, same L1): s_waitcnt vmcnt(0) is all the hardware needs */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* unaligned little-endian dword at q (wave-uniform), served from the window */
__device__ __forceinline__ uint32_t win_u32(src_window &W, const uint8_t *q, int lane)
{
	uint32_t d = (uint32_t)(q - W.base);
	if (d + 4 > 256) {
		if (d < 256) {
			/* straddles the two halves: assemble from bytes (rare) */
			uint32_t v = 0;
			for (int k = 0; k < 4; k++)
				v |= win_byte(W, q + k, lane) << (8 * k);
			return v;
		}
		win_byte(W, q, lane);	/* slides or reloads the window */
		d = (uint32_t)(q - W.base);
	}
	uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)W.w0, (int)(d >> 2));
	uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)W.w0, (int)(((d >> 2) + 1) & 63));
	if ((d >> 2) == 63)
		hi = (uint32_t)__builtin_amdgcn_readlane((int)W.w1, 0);
	return __builtin_amdgcn_alignbyte(hi, lo, d & 3);
}

/* LDS accepts unaligned 2/4/8-byte accesses on gfx950 (the compiler emits
 * ds_read_b64 / ds_write_b64 for these), so window copies move 8 bytes per
 * instruction at any byte address. */
__device__ __forceinline__ uint64_t lds_ld8(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void lds_st8(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void lds_st4(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ uint4 lds_ld16(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void lds_st16(uint8_t *p, uint4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ void lds_st2(uint8_t *p, uint16_t v) { __builtin_memcpy(p, &v, 2); }

/* store the low n (< 8) bytes of v */
__device__ __forceinline__ void lds_st_tail(uint8_t *p, uint64_t v, uint32_t n)
{
	if (n & 4) { lds_st4(p, (uint32_t)v); p += 4; v >>= 32; }
	if (n & 2) { lds_st2(p, (uint16_t)v); p += 2; v >>= 16; }
	if (n & 1) *p = (uint8_t)v;
}

/* load n (< 8) bytes without touching bytes past p+n */
__device__ __forceinline__ uint64_t lds_ld_tail(const uint8_t *p, uint32_t n)
{
	uint64_t v = 0;
	uint32_t sh = 0;
	if (n & 4) { uint32_t t; __builtin_memcpy(&t, p, 4); v = t; p += 4; sh = 32; }
	if (n & 2) { uint16_t t; __builtin_memcpy(&t, p, 2); v |= (uint64_t)t << sh; p += 2; sh += 16; }
	if (n & 1) v |= (uint64_t)(*p) << sh;
	return v;
}

/* The usual match -- at most 64 bytes, not overlapping its source -- as ONE straight line of predicated DS
 * instructions.  A wave on its own issues an instruction every four or five cycles and pays an instruction-fetch
 * restart for every taken branch, and the matcher's passes are nothing but this copy: what the compiler makes of
 * `if (class) copy piece` is three to eight instructions and up to two branches per piece.  Here a piece is two:
 * exec = ready lanes of the class (both wave-uniform masks, the class masks worked out once per group), then the
 * DS instruction; no branch.  All loads are issued before anything is stored, pieces as in io_copy_exact:
 * 16-byte pieces at 0 / 16 / 32 and one that ENDS with the match, 8 + 8 overlapping below 16 bytes, 4 + 4 below 8,
 * 2 + 1 below 4 (1..3 bytes: only the deflate front end makes these). */
typedef uint32_t io_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t io_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint8_t io_lds_u8;
__device__ __forceinline__ uint32_t io_lds_addr(const uint8_t *p) { return (uint32_t)(uintptr_t)(const io_lds_u8 *)p; }

struct io_cls {		/* lanes of the group by length class, fast lanes only (wave-uniform, one SGPR pair each) */
	uint64_t k16, k32, k48, ke;	/* >= 16, >= 32, >= 48, end piece (length > 16 and not 32 or 48) */
	uint64_t ks, ks8, ks88;		/* < 16 (8-byte load), 8..15, 9..15 */
	uint64_t ks4, ks44, k2, k1;	/* 4..7, 5..7, {2,3}, {1,3} */
};
struct io_adr {		/* per lane: LDS byte addresses and shift counts */
	uint32_t fp, mp;	/* source, destination */
	uint32_t fe, me;	/* + length - 16 */
	uint32_t f8, m8;	/* + length - 8 */
	uint32_t m4, sh4;	/* mp + length - 4, 8 * (length - 4) */
	uint32_t m1, sh1;	/* mp + (length & 2), 8 * (length & 2) */
};

/* one predicated DS instruction: exec = ready lanes of the class, then the instruction (IO_EXP_SKIP_EMPTY: the
 * instruction is jumped over when no lane is left -- a timing experiment, see profiles/r03_inorder_whatif.txt) */
#ifdef IO_EXP_SKIP_EMPTY
#define IO_DS(mask_, ins_, n_) "s_and_b64 exec, %[R], %[" mask_ "]\n\ts_cbranch_execz .Lio" n_ "_%=\n\t" ins_ "\n\t.Lio" n_ "_%=:\n\t"
#else
#define IO_DS(mask_, ins_, n_) "s_and_b64 exec, %[R], %[" mask_ "]\n\t" ins_ "\n\t"
#endif
__device__ __forceinline__ void io_copy_fast(const uint64_t R, const io_cls &K, const io_adr &A)
{
	io_u32x4 v0, v1, v2, vt;
	io_u32x2 a0, at;
	uint64_t sv;
	asm volatile(
	    "s_mov_b64 %[sv], exec\n\t"
	    IO_DS("k16", "ds_read_b128 %[v0], %[fp]", "0")
	    IO_DS("k32", "ds_read_b128 %[v1], %[fp] offset:16", "1")
	    IO_DS("k48", "ds_read_b128 %[v2], %[fp] offset:32", "2")
	    IO_DS("ke", "ds_read_b128 %[vt], %[fe]", "3")
	    IO_DS("ks", "ds_read_b64 %[a0], %[fp]", "4")
	    IO_DS("ks88", "ds_read_b64 %[at], %[f8]", "5")
	    "s_mov_b64 exec, %[sv]\n\t"
	    "s_waitcnt lgkmcnt(0)"
	    : [sv] "=&s"(sv), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [vt] "=&v"(vt), [a0] "=&v"(a0), [at] "=&v"(at)
	    : [R] "s"(R), [k16] "s"(K.k16), [k32] "s"(K.k32), [k48] "s"(K.k48), [ke] "s"(K.ke), [ks] "s"(K.ks), [ks88] "s"(K.ks88),
	      [fp] "v"(A.fp), [fe] "v"(A.fe), [f8] "v"(A.f8)
	    : "memory");
	/* (lanes outside a class hold garbage in that class's registers: never stored) */
#ifdef IO_EXP_LOADS_ONLY	/* timing experiment */
	asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(vt), "v"(a0), "v"(at));
	return;
#endif
	const uint32_t a0lo = a0.x;
	const uint32_t t4 = (uint32_t)((((uint64_t)a0.y << 32) | a0.x) >> (A.sh4 & 63u));
	const uint32_t t1 = a0lo >> (A.sh1 & 31u);
	asm volatile(
	    "s_mov_b64 %[sv], exec\n\t"
	    IO_DS("k16", "ds_write_b128 %[mp], %[v0]", "0")
	    IO_DS("k32", "ds_write_b128 %[mp], %[v1] offset:16", "1")
	    IO_DS("k48", "ds_write_b128 %[mp], %[v2] offset:32", "2")
	    IO_DS("ke", "ds_write_b128 %[me], %[vt]", "3")
	    IO_DS("ks8", "ds_write_b64 %[mp], %[a0]", "4")
	    IO_DS("ks88", "ds_write_b64 %[m8], %[at]", "5")
	    IO_DS("ks4", "ds_write_b32 %[mp], %[a0lo]", "6")
	    IO_DS("ks44", "ds_write_b32 %[m4], %[t4]", "7")
	    IO_DS("k2", "ds_write_b16 %[mp], %[a0lo]", "8")
	    IO_DS("k1", "ds_write_b8 %[m1], %[t1]", "9")
	    "s_mov_b64 exec, %[sv]"
	    : [sv] "=&s"(sv)
	    : [R] "s"(R), [k16] "s"(K.k16), [k32] "s"(K.k32), [k48] "s"(K.k48), [ke] "s"(K.ke), [ks8] "s"(K.ks8), [ks88] "s"(K.ks88),
	      [ks4] "s"(K.ks4), [ks44] "s"(K.ks44), [k2] "s"(K.k2), [k1] "s"(K.k1),
	      [mp] "v"(A.mp), [me] "v"(A.me), [m8] "v"(A.m8), [m4] "v"(A.m4), [m1] "v"(A.m1),
	      [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [vt] "v"(vt), [a0] "v"(a0), [at] "v"(at), [a0lo] "v"(a0lo), [t4] "v"(t4), [t1] "v"(t1)
	    : "memory");
}
#undef IO_DS

/* ---- launch interface (definitions live next to their kernels) ---- */

/* la_hash.hip */
void la_launch_xxh32_many(hipStream_t s, const uint8_t *d_base, const la_hash_job *d_jobs,
    uint32_t n, uint32_t *d_out);
void la_launch_crc32_many(hipStream_t s, const uint8_t *d_base, const la_hash_job *d_jobs,
    uint32_t n, uint32_t *d_out);
void la_launch_lz4_block_sums(hipStream_t s, const uint8_t *d_src, const la_lz4_block *d_blocks,
    uint32_t n, uint32_t *d_status);
void la_launch_lz4_frame_sums(hipStream_t s, const uint8_t *d_src, const uint8_t *d_dst,
    const la_lz4_frame *d_frames, uint32_t n_frames, const uint64_t *d_dst_off,
    uint64_t dst_cap, uint32_t *d_frame_status, uint32_t end_lo, uint32_t end_hi,
    const void *d_carry_in, void *d_carry_out, int row /* 16 lanes per frame: latency over throughput */);
void la_launch_lz4_merge_status(hipStream_t s, const uint32_t *d_sum_status, uint32_t n, uint32_t *d_status);

/* la_lz4.hip, la_lz4_fast.hip */
struct la_lz4_seq {	/* one LZ4 sequence, 8 bytes, written by the parse kernel */
	uint16_t lit_src;	/* offset of the first literal inside the block payload */
	uint16_t lit_len;
	uint16_t dst;		/* output offset where the literals go */
	uint16_t off;		/* match offset (0 in a final, literal-only sequence) */
};
/* match length of sequence k = (k+1 < nseq ? seq[k+1].dst : out_len) - (dst + lit_len) */

#define LA_LZ4_FAST_MAXSEQ 4096u	/* sequences per block the LDS-window kernel holds (per segment) */
#define LA_INFLATE_MAXSEQ 12288u	/* table entries per deflate member in the two-phase path (three segments) */

/* blocks the LDS-window kernel may take: compressed, independent, window <= 64 KiB */
__host__ __device__ __forceinline__ bool la_lz4_fast_eligible(const la_lz4_block &b)
{
	return !(b.flags & (LA_LZ4B_STORED | LA_LZ4B_DEPENDENT)) && b.dst_cap <= 65536u && b.src_len <= 65536u;
}

/* Blocks made of few, long sequences (long runs, big repeats: on average LA_LZ4_LONG_SEQ_BYTES decoded bytes per
 * sequence or more) go to the wave-wide general kernel even when the LDS-window kernels could take them: one lane per
 * match copies a 64 KiB run at the speed of one lane (10 GB/s on blocks of zeros), the general kernel's wave-wide
 * copies reach 250 GB/s on them (tools/measure_long_matches.py).  thr = 0 switches the routing off (the deflate
 * front end has no general kernel behind it).  The same predicate is used on both sides. */
#define LA_LZ4_LONG_SEQ_BYTES 512u
__host__ __device__ __forceinline__ bool la_lz4_long_sequences(uint32_t nseq, uint32_t out_len, uint32_t thr)
{
	return thr != 0 && nseq != 0xFFFFFFFFu && (uint64_t)nseq * thr <= out_len;
}

void la_launch_lz4_table_caps(hipStream_t s, const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_caps);
void la_launch_lz4_parse(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_out_len, uint32_t *d_nseq,
    uint32_t *d_status, la_lz4_seq *d_table /* NULL: measure only */, const uint64_t *d_table_off,
    uint64_t table_cap /* entries */);
/* la_lz4_parse.hip: the same parse fed from an LDS-staged image, block checksums fused in
 * (d_sum_status NULL: no checksums) */
void la_launch_lz4_parse_staged(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_out_len, uint32_t *d_nseq,
    uint32_t *d_status, uint32_t *d_sum_status, la_lz4_seq *d_table, const uint64_t *d_table_off,
    uint64_t table_cap);
void la_launch_lz4_expand_general(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, const uint32_t *d_status,
    const uint32_t *d_nseq, uint32_t fast_max_seq /* 0: take every block */, uint32_t hist_len, uint32_t long_thr);
void la_launch_lz4_expand_fast(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t long_thr);
void la_launch_lz4_expand_fast_big(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off,
    uint32_t *d_big /* n + 1 words */, uint32_t long_thr);

/* la_lz4_inorder.hip: the default expand step since round 3 (in-order matcher wave + literal wave + flush wave per
 * 64 KiB LDS window); the same contract, blocks of any sequence count (no _big launch) */
bool la_lz4_expand_inorder_takes(uint64_t src_bytes);	/* false: image too short for it, use the polling kernel */
void la_launch_lz4_expand_inorder(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off, uint32_t long_thr);

/* la_zstd.hip */
uint64_t la_zstd_workspace_bytes(uint32_t n_frames);
void la_launch_zstd_frames(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes, const la_zstd_frame *d_frames, uint32_t n,
    uint8_t *d_dst, uint64_t dst_cap, la_zstd_result *d_results, uint8_t *ws, uint32_t options);

/* la_lz4_comp.hip */
void la_launch_lz4_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes, uint32_t block_size,
    uint32_t bpf, uint32_t flags, uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_bytes, uint8_t *ws);

/* la_deflate_comp.hip */
void la_launch_gzip_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes, uint32_t chunk, uint32_t mtime,
    uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_bytes, uint8_t *ws);

/* la_inflate.hip */
void la_launch_inflate(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, uint8_t *d_dst, uint64_t dst_cap, la_gz_result *d_results);
uint64_t la_inflate_lanes_scratch_bytes(uint32_t n);
/* outputs of the entropy-only launch (la_launch_inflate_symbols): everything
 * lz4_expand_fast_kernel needs to build the members in its LDS window */
struct la_inflate_emit {
	uint8_t *lit;		/* literal bytes, 64 KiB per member */
	la_lz4_seq *table;	/* LA_INFLATE_MAXSEQ entries per member */
	la_lz4_block *blocks;	/* [n] literal buffer as the expand kernel's "payload" */
	uint32_t *out_len;	/* [n] */
	uint32_t *nseq;		/* [n] */
	uint32_t *xstatus;	/* [n] 0 = expand it, else skip (member goes to the in-place kernel) */
	uint32_t *todo;		/* [n] 1 = the in-place kernel must decode this member */
	uint64_t *dst_off;	/* [n+1] */
	uint64_t *table_off;	/* [n+1] */
};
void la_launch_inflate_lanes(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, uint8_t *d_dst, uint64_t dst_cap, la_gz_result *d_results,
    void *d_scratch, const uint32_t *d_only /* NULL: every member */);
void la_launch_inflate_symbols(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, uint64_t dst_cap, la_gz_result *d_results,
    void *d_scratch, la_inflate_emit E);
void la_launch_gz_verify(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, const uint8_t *d_dst, la_gz_result *d_results, int verify);
void la_launch_gz_summary(hipStream_t s, const la_gz_result *d_results, uint32_t n, la_batch_summary *d_summary);

/* la_scan.hip */
void la_launch_scan_u32(hipStream_t s, const uint32_t *d_in, uint32_t n, uint64_t *d_out /* n+1 */,
    void *d_scratch);
void la_launch_scan_u32_base(hipStream_t s, const uint32_t *d_in, uint32_t n, uint64_t *d_out, void *d_scratch,
    const uint64_t *d_base);
uint64_t la_scan_scratch_bytes(uint32_t n);
void la_launch_lz4_summary(hipStream_t s, const uint32_t *d_out_len, const uint32_t *d_block_status,
    uint32_t n_blocks, const uint32_t *d_frame_status, uint32_t n_frames,
    const uint64_t *d_dst_off, la_batch_summary *d_summary);

#endif
