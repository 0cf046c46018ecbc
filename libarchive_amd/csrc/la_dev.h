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
/* one whole hash computed by ONE lane (many-hash kernels run one hash per lane) */
__device__ __forceinline__ uint32_t xxh32_lane(const uint8_t *p, uint32_t len, uint32_t seed)
{
	const uint8_t *end = p + len;
	uint32_t h;
	if (len >= 16) {
		uint32_t v1 = seed + XXH_P1 + XXH_P2, v2 = seed + XXH_P2, v3 = seed, v4 = seed - XXH_P1;
		const uint8_t *limit = end - 16;
		do {
			uint4 x = ld_u128(p);
			v1 = xxh_round(v1, x.x);
			v2 = xxh_round(v2, x.y);
			v3 = xxh_round(v3, x.z);
			v4 = xxh_round(v4, x.w);
			p += 16;
		} while (p <= limit);
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
    uint64_t dst_cap, uint32_t *d_frame_status);

/* la_lz4.hip, la_lz4_fast.hip */
struct la_lz4_seq {	/* one LZ4 sequence, 8 bytes, written by the parse kernel */
	uint16_t lit_src;	/* offset of the first literal inside the block payload */
	uint16_t lit_len;
	uint16_t dst;		/* output offset where the literals go */
	uint16_t off;		/* match offset (0 in a final, literal-only sequence) */
};
/* match length of sequence k = (k+1 < nseq ? seq[k+1].dst : out_len) - (dst + lit_len) */

#define LA_LZ4_FAST_MAXSEQ 4096u	/* sequences per block the LDS-window kernel holds */

/* blocks the LDS-window kernel may take: compressed, independent, window <= 64 KiB */
__host__ __device__ __forceinline__ bool la_lz4_fast_eligible(const la_lz4_block &b)
{
	return !(b.flags & (LA_LZ4B_STORED | LA_LZ4B_DEPENDENT)) && b.dst_cap <= 65536u && b.src_len <= 65536u;
}

void la_launch_lz4_table_caps(hipStream_t s, const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_caps);
void la_launch_lz4_parse(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_out_len, uint32_t *d_nseq,
    uint32_t *d_status, la_lz4_seq *d_table /* NULL: measure only */, const uint64_t *d_table_off,
    uint64_t table_cap /* entries */);
void la_launch_lz4_expand_general(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, const uint32_t *d_status,
    const uint32_t *d_nseq, uint32_t fast_max_seq /* 0: take every block */);
void la_launch_lz4_expand_fast(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, const uint32_t *d_status,
    const uint32_t *d_nseq, const la_lz4_seq *d_table, const uint64_t *d_table_off);

/* la_scan.hip */
void la_launch_scan_u32(hipStream_t s, const uint32_t *d_in, uint32_t n, uint64_t *d_out /* n+1 */,
    void *d_scratch);
uint64_t la_scan_scratch_bytes(uint32_t n);
void la_launch_lz4_summary(hipStream_t s, const uint32_t *d_out_len, const uint32_t *d_block_status,
    uint32_t n_blocks, const uint32_t *d_frame_status, uint32_t n_frames,
    const uint64_t *d_dst_off, la_batch_summary *d_summary);

#endif
