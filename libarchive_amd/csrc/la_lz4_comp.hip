/*
 * la_lz4_comp.hip -- LZ4 block COMPRESSION + frame assembly on the device (gfx950): the data
 * plane of the lz4 write filter.
 *
 * Replaces, for a whole stream per call, what libarchive/archive_write_add_filter_lz4.c does
 * per block on the host: LZ4_compress_default of one independent block (:484-532,
 * drive_compressor_independence), the stored-block fallback when a block does not shrink
 * (:505-515), the block checksum XXH32 over the bytes as written (:516-521), the frame
 * descriptor with its check byte (:394-419), the EndMark and the content checksum (:433-447).
 * The compressed bytes are NOT those of liblz4 (an LZ4 stream is not unique): parity for this
 * direction is the round trip -- what the reference's reader, the oracle and liblz4 decode from
 * these frames must be the input, bit for bit -- plus the format rules every decoder checks
 * (last five bytes literal, last match at least 12 bytes before the end, offsets inside the block).
 *
 *   lz4_compress_blocks_kernel   ONE WAVE per block of at most 64 KiB.  A 4096-entry hash table of
 *       16-bit positions in LDS; the wave looks at 64 consecutive positions at a time: every lane
 *       hashes the four bytes at its position, takes the table's candidate (from an earlier window),
 *       replaces it, verifies the candidate and extends the match eight bytes at a time.  The wave
 *       then takes the matches in position order (ballot + first set bit), skipping the ones an
 *       earlier match has covered, and writes each sequence cooperatively: token and length bytes by
 *       the first lanes, the literal run 64 bytes per step.  Input is read from global memory
 *       (it stays in L1 / L2 for the lifetime of a block), so LDS holds only the table and many
 *       waves fit a CU.
 *   lz4c_sizes_kernel / scan     stream bytes each block and frame contributes -> offsets.
 *   lz4_pack_frames_kernel       one workgroup per block: size word, payload (compressed, or the
 *       input itself when it did not shrink), block checksum; the frame's first / last block also
 *       writes the 7-byte header / EndMark + content checksum.
 * The two kinds of XXH32 (per block over the written payload, per frame over the input) run on
 * xxh32_lane / xxh32_quad of la_dev.h.
 */
#include "la_dev.h"

#define LZ4C_HASH_BITS 12
#define LZ4C_MFLIMIT   12u
#define LZ4C_LASTLIT   5u

__host__ __device__ static inline uint32_t lz4c_bound(uint32_t n) { return n + n / 255u + 16u; }

__device__ __forceinline__ uint64_t ld_u64(const uint8_t *p)
{
	uint64_t v;
	__builtin_memcpy(&v, p, 8);
	return v;
}

/* wave-cooperative byte copy, n uniform */
__device__ __forceinline__ void wave_copy(uint8_t *d, const uint8_t *s, uint32_t n, uint32_t lane)
{
	for (uint32_t i = lane; i < n; i += 64)
		d[i] = s[i];
}

/* length field beyond the token nibble: v - 15 as 255, 255, ..., rest (lanes write in parallel); returns bytes written */
__device__ __forceinline__ uint32_t wave_put_len(uint8_t *d, uint32_t v, uint32_t lane)
{
	const uint32_t rest = v - 15u, cnt = rest / 255u + 1u;
	for (uint32_t i = lane; i < cnt; i += 64)
		d[i] = (i + 1u < cnt) ? (uint8_t)255 : (uint8_t)(rest - 255u * (cnt - 1u));
	return cnt;
}

__global__ __launch_bounds__(64) void lz4_compress_blocks_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    uint32_t block_size, uint32_t n_blocks, uint8_t *__restrict__ tmp, uint32_t tmp_stride,
    uint32_t *__restrict__ csize)
{
	__shared__ uint16_t tab[1u << LZ4C_HASH_BITS];
	const uint32_t bi = blockIdx.x, lane = threadIdx.x;
	if (bi >= n_blocks)
		return;
	const uint64_t so = (uint64_t)bi * block_size;
	const uint32_t n = (uint32_t)(src_bytes - so < block_size ? src_bytes - so : block_size);
	const uint8_t *in = src + so;
	uint8_t *out = tmp + (uint64_t)bi * tmp_stride;
	for (uint32_t i = lane; i < (1u << LZ4C_HASH_BITS); i += 64)
		tab[i] = 0;
	__syncthreads();

	uint32_t anchor = 0, op = 0;	/* wave-uniform */
	if (n > LZ4C_MFLIMIT) {
		const uint32_t mflimit = n - LZ4C_MFLIMIT;	/* last position a match may START at */
		const uint32_t matchlimit = n - LZ4C_LASTLIT;	/* matches END at or before this */
		uint32_t base = 0;
		while (base <= mflimit) {
			const uint32_t p = base + lane;
			const bool valid = p <= mflimit;
			uint32_t v = 0, cand = 0, mlen = 0;
			bool ok = false;
			if (valid) {
				v = ld_u32(in + p);
				const uint32_t h = (v * 2654435761u) >> (32 - LZ4C_HASH_BITS);
				cand = tab[h];		/* every lane reads before any lane of this window writes */
			}
			__builtin_amdgcn_wave_barrier();
			if (valid) {
				const uint32_t h = (v * 2654435761u) >> (32 - LZ4C_HASH_BITS);
				tab[h] = (uint16_t)p;
				/* (position 0 doubles as "empty": a candidate is only taken if its bytes match) */
				ok = cand < p && ld_u32(in + cand) == v;
				if (ok) {
					mlen = 4;
					while (p + mlen + 8 <= matchlimit && ld_u64(in + p + mlen) == ld_u64(in + cand + mlen))
						mlen += 8;
					while (p + mlen < matchlimit && in[p + mlen] == in[cand + mlen])
						mlen++;
				}
			}
			uint64_t mask = __ballot(ok);
			while (mask != 0) {
				const uint32_t f = (uint32_t)__builtin_ctzll(mask);
				mask &= mask - 1;
				const uint32_t pf = base + f;
				if (pf < anchor)
					continue;	/* an earlier match of this window already covers it */
				const uint32_t mf = (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)f);
				const uint32_t cf = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)f);
				const uint32_t lit = pf - anchor, off = pf - cf, ml = mf - 4u;
				/* one sequence: token, literal length, literals, offset, match length */
				if (lane == 0)
					out[op] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (ml < 15u ? ml : 15u));
				op += 1;
				if (lit >= 15u)
					op += wave_put_len(out + op, lit, lane);
				wave_copy(out + op, in + anchor, lit, lane);
				op += lit;
				if (lane == 0) {
					out[op] = (uint8_t)off;
					out[op + 1] = (uint8_t)(off >> 8);
				}
				op += 2;
				if (ml >= 15u)
					op += wave_put_len(out + op, ml, lane);
				anchor = pf + mf;
			}
			base = (base + 64 > anchor) ? base + 64 : anchor;
		}
	}
	/* last sequence: literals only (at least the block's last five bytes) */
	{
		const uint32_t lit = n - anchor;
		if (lane == 0)
			out[op] = (uint8_t)((lit < 15u ? lit : 15u) << 4);
		op += 1;
		if (lit >= 15u)
			op += wave_put_len(out + op, lit, lane);
		wave_copy(out + op, in + anchor, lit, lane);
		op += lit;
	}
	if (lane == 0)
		csize[bi] = op;
}

/* bytes of the stream each block contributes: size word + payload (stored when it did not shrink) + block sum,
 * plus the frame's header in front of its first block and EndMark (+ content sum) behind its last */
__global__ __launch_bounds__(256) void lz4c_sizes_kernel(const uint32_t *__restrict__ csize, uint64_t src_bytes,
    uint32_t block_size, uint32_t n_blocks, uint32_t bpf, uint32_t flags, uint32_t *__restrict__ contrib)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_blocks)
		return;
	const uint64_t so = (uint64_t)i * block_size;
	const uint32_t n = (uint32_t)(src_bytes - so < block_size ? src_bytes - so : block_size);
	const uint32_t pay = csize[i] < n ? csize[i] : n;
	uint32_t c = 4u + pay + ((flags & LA_LZ4C_BLOCK_SUM) ? 4u : 0u);
	if (i % bpf == 0)
		c += 7u;
	if (i % bpf == bpf - 1 || i + 1 == n_blocks)
		c += 4u + ((flags & LA_LZ4C_CONTENT_SUM) ? 4u : 0u);
	contrib[i] = c;
}

__device__ __forceinline__ void st_le32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

__global__ __launch_bounds__(256) void lz4_pack_frames_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    uint32_t block_size, uint32_t n_blocks, uint32_t bpf, uint32_t flags, const uint8_t *__restrict__ tmp,
    uint32_t tmp_stride, const uint32_t *__restrict__ csize, const uint64_t *__restrict__ off,
    const uint32_t *__restrict__ frame_sum, uint8_t *__restrict__ out, uint64_t out_cap, uint64_t *__restrict__ out_bytes)
{
	const uint32_t bi = blockIdx.x, tid = threadIdx.x;
	if (bi >= n_blocks)
		return;
	const uint64_t so = (uint64_t)bi * block_size;
	const uint32_t n = (uint32_t)(src_bytes - so < block_size ? src_bytes - so : block_size);
	const bool stored = csize[bi] >= n;
	const uint32_t pay = stored ? n : csize[bi];
	const uint8_t *payload = stored ? src + so : tmp + (uint64_t)bi * tmp_stride;
	uint64_t o = off[bi];
	if (bi + 1 == n_blocks && tid == 0)
		*out_bytes = off[n_blocks];
	if (off[bi + 1] > out_cap)
		return;		/* the caller sees out_bytes > out_cap */
	__shared__ uint32_t bsum;
	if (bi % bpf == 0) {
		if (tid == 0) {
			/* magic, FLG (version 01, independent blocks, checksums as asked), BD (block maximum), check byte */
			const uint8_t flg = (uint8_t)(0x60 | ((flags & LA_LZ4C_BLOCK_SUM) ? 0x10 : 0) | ((flags & LA_LZ4C_CONTENT_SUM) ? 0x04 : 0));
			const uint8_t bd = block_size <= 65536u ? 0x40 : block_size <= 262144u ? 0x50 : block_size <= 1048576u ? 0x60 : 0x70;
			st_le32(out + o, 0x184D2204u);
			out[o + 4] = flg;
			out[o + 5] = bd;
			const uint8_t d[2] = { flg, bd };
			out[o + 6] = (uint8_t)(xxh32_lane(d, 2, 0) >> 8);
		}
		o += 7;
	}
	if (tid == 0)
		st_le32(out + o, stored ? (pay | 0x80000000u) : pay);
	o += 4;
	for (uint32_t i = tid; i < pay; i += 256)
		out[o + i] = payload[i];
	if (flags & LA_LZ4C_BLOCK_SUM) {
		/* over the payload as written (archive_write_add_filter_lz4.c:516-521); four lanes share the chain */
		if (tid < 4) {
			const uint32_t h = xxh32_quad(payload, pay, 0, (int)tid);
			if (tid == 0)
				bsum = h;
		}
		__syncthreads();
		if (tid == 0)
			st_le32(out + o + pay, bsum);
	}
	o += pay + ((flags & LA_LZ4C_BLOCK_SUM) ? 4u : 0u);
	if ((bi % bpf == bpf - 1 || bi + 1 == n_blocks) && tid == 0) {
		st_le32(out + o, 0);	/* EndMark */
		if (flags & LA_LZ4C_CONTENT_SUM)
			st_le32(out + o + 4, frame_sum[bi / bpf]);
	}
}

/* content checksum of every frame: XXH32 over its input bytes, four lanes per frame */
__global__ __launch_bounds__(64) void lz4c_frame_sums_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    uint32_t block_size, uint32_t bpf, uint32_t n_frames, uint32_t *__restrict__ frame_sum)
{
	const uint32_t q = (blockIdx.x * 64 + threadIdx.x) >> 2, j = threadIdx.x & 3;
	const bool have = q < n_frames;
	const uint64_t fo = (uint64_t)(have ? q : 0) * bpf * block_size;
	const uint64_t fl = have ? (src_bytes - fo < (uint64_t)bpf * block_size ? src_bytes - fo : (uint64_t)bpf * block_size) : 0;
	const uint32_t h = xxh32_quad(src + fo, (uint32_t)fl, 0, (int)j);
	if (have && j == 0)
		frame_sum[q] = h;
}

extern "C" uint64_t la_gpu_lz4_compress_workspace_bytes(uint64_t src_bytes, uint32_t block_size, uint32_t blocks_per_frame)
{
	if (block_size == 0 || blocks_per_frame == 0)
		return 0;
	const uint64_t nb = (src_bytes + block_size - 1) / block_size, nf = (nb + blocks_per_frame - 1) / blocks_per_frame;
	const uint64_t stride = (lz4c_bound(block_size) + 15u) & ~15ull;
	return nb * stride + nb * 4 * 2 + (nb + 1) * 8 + nf * 4 + la_scan_scratch_bytes((uint32_t)nb) + 4096;
}

void la_launch_lz4_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes, uint32_t block_size,
    uint32_t bpf, uint32_t flags, uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_bytes, uint8_t *ws)
{
	const uint32_t nb = (uint32_t)((src_bytes + block_size - 1) / block_size);
	const uint32_t nf = (nb + bpf - 1) / bpf;
	const uint32_t stride = (lz4c_bound(block_size) + 15u) & ~15u;
	uint64_t o = 0;
	uint8_t *tmp = ws + o; o += (uint64_t)nb * stride;
	uint32_t *csize = (uint32_t *)(ws + o); o += (uint64_t)nb * 4;
	uint32_t *contrib = (uint32_t *)(ws + o); o += (uint64_t)nb * 4;
	o = (o + 7) & ~7ull;
	uint64_t *off = (uint64_t *)(ws + o); o += ((uint64_t)nb + 1) * 8;
	uint32_t *fsum = (uint32_t *)(ws + o); o += (uint64_t)nf * 4;
	o = (o + 255) & ~255ull;
	void *scan = ws + o;
	if (nb == 0) {
		(void)hipMemsetAsync(d_out_bytes, 0, 8, s);
		return;
	}
	hipLaunchKernelGGL(lz4_compress_blocks_kernel, dim3(nb), dim3(64), 0, s, d_src, src_bytes, block_size, nb, tmp, stride, csize);
	if (flags & LA_LZ4C_CONTENT_SUM)
		hipLaunchKernelGGL(lz4c_frame_sums_kernel, dim3((nf + 15) / 16), dim3(64), 0, s, d_src, src_bytes, block_size, bpf, nf, fsum);
	hipLaunchKernelGGL(lz4c_sizes_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, csize, src_bytes, block_size, nb, bpf, flags, contrib);
	la_launch_scan_u32(s, contrib, nb, off, scan);
	hipLaunchKernelGGL(lz4_pack_frames_kernel, dim3(nb), dim3(256), 0, s, d_src, src_bytes, block_size, nb, bpf, flags,
	    tmp, stride, csize, off, fsum, d_out, out_cap, d_out_bytes);
}
