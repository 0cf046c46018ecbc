/*
 * la_hash.hip -- many-hash XXH32 and wave-reduced CRC32 kernels (gfx950).
 *
 * Replaces, for whole batches, the per-call hashing of
 *   libarchive/xxhash.c:234-319        (XXH32 one shot; call sites lz4.c:446, :518, :652)
 *   libarchive/archive_crc32.h:43-84   (crc32; the gzip trailer value the reference
 *                                       never checks, gzip.c:423)
 * Integer/byte work bound by HBM reads of the hashed bytes; no MFMA.
 *
 * XXH32 is a serial chain per hash, so the parallel axis is "many
 * hashes": one lane owns one hash and streams its range with 16-byte loads.
 * CRC32 is GF(2)-linear, so ONE range is split over the 64 lanes of a wave
 * (contiguous chunks, slicing-by-4 tables in LDS) and the lane values are
 * merged with a log2(64)-step shuffle tree of carry-less multiplications by
 * x^(8*chunk*2^k) mod P.
 */
#include "la_dev.h"

/* ------------------------------------------------------------------ XXH32 */

__global__ __launch_bounds__(256) void xxh32_many_kernel(const uint8_t *__restrict__ base,
    const la_hash_job *__restrict__ jobs, uint32_t n, uint32_t *__restrict__ out)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	la_hash_job j = jobs[i];
	out[i] = xxh32_lane(base + j.off, j.len, j.seed);
}

/* checksum verdicts outrank decode verdicts (the reference checks the block checksum
 * first, lz4.c:517 vs :594): fold them into the per-block status */
__global__ __launch_bounds__(256) void lz4_merge_status_kernel(const uint32_t *__restrict__ sum_status,
    uint32_t n, uint32_t *__restrict__ status)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && sum_status[i] != LA_ST_OK)
		status[i] = sum_status[i];
}

void la_launch_lz4_merge_status(hipStream_t s, const uint32_t *d_sum_status, uint32_t n, uint32_t *d_status)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_merge_status_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_sum_status, n, d_status);
}

/* block checksums: XXH32 over the compressed payload (lz4.c:517-526) */
__global__ __launch_bounds__(256) void lz4_block_sums_kernel(const uint8_t *__restrict__ src,
    const la_lz4_block *__restrict__ blocks, uint32_t n, uint32_t *__restrict__ status)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	la_lz4_block b = blocks[i];
	if (!(b.flags & LA_LZ4B_CHECKSUM))
		return;
	uint32_t h = xxh32_lane(src + b.src_off, b.src_len, 0);
	if (h != b.block_sum)
		status[i] = LA_ST_LZ4_BAD_BLOCK_SUM;	/* outranks a decode failure: checked first, lz4.c:517 vs :594 */
}

/* per frame: descriptor check byte (lz4.c:446-451) and content checksum over the
 * frame's decoded bytes (lz4.c:639-665).  Four lanes per frame (xxh32_quad): frames
 * are few and their chains long. */
/* XXH32 state of a content checksum that spans batches (LA_XXH_CARRY_BYTES on the device):
 * the streaming form of libarchive/xxhash.c:325-507 (XXH32_init / _update / _digest) */
struct la_xxh_carry {
	uint32_t v[4];
	uint32_t memsize;	/* bytes waiting in mem for a full 16-byte stripe */
	uint32_t total_lo, total_hi;
	uint8_t mem[16];
	uint32_t pad[5];
};
static_assert(sizeof(la_xxh_carry) == LA_XXH_CARRY_BYTES, "carry record size");

/*
 * One piece [p, p+n) of a hash that began in an earlier batch (cin != NULL) and/or goes on in
 * the next one (cout != NULL).  Four lanes per hash as in xxh32_quad: lane j owns accumulator
 * j.  Returns the digest when cout == NULL (valid in all four lanes).
 */
/* ROW: called by the sixteen lanes of a DPP row (l = 0..15, accumulator j = l & 3 valid in lanes
 * 0..3) -- the stripes are then chained as in xxh32_row; otherwise by the four lanes of a quad. */
template <bool ROW>
__device__ uint32_t xxh32_quad_stream(const la_xxh_carry *cin, la_xxh_carry *cout, const uint8_t *p, uint64_t n, int l)
{
	const int j = l & 3;
	uint32_t v = (j == 0) ? XXH_P1 + XXH_P2 : (j == 1) ? XXH_P2 : (j == 2) ? 0u : 0u - XXH_P1;
	uint32_t msz = 0;
	uint64_t total = 0;
	uint8_t mem[16];
	for (int k = 0; k < 16; k++) mem[k] = 0;
	if (cin) {
		v = cin->v[j];
		msz = cin->memsize;
		total = ((uint64_t)cin->total_hi << 32) | cin->total_lo;
		for (int k = 0; k < 16; k++) mem[k] = cin->mem[k];
	}
	total += n;
	if (msz + n < 16) {
		/* still no full stripe: the bytes join the waiting ones (xxhash.c:377-381) */
		for (uint64_t k = 0; k < n; k++) mem[msz + k] = p[k];
		msz += (uint32_t)n;
		n = 0;
	} else {
		if (msz) {
			/* complete the waiting stripe first (xxhash.c:383-401) */
			const uint32_t need = 16 - msz;
			for (uint32_t k = 0; k < need; k++) mem[msz + k] = p[k];
			const uint32_t dw = (uint32_t)mem[4 * j] | ((uint32_t)mem[4 * j + 1] << 8) |
			    ((uint32_t)mem[4 * j + 2] << 16) | ((uint32_t)mem[4 * j + 3] << 24);
			v = xxh_round(v, dw);
			p += need; n -= need; msz = 0;
		}
		const uint8_t *q = p + 4 * j;
		uint64_t left = n;
		if (ROW) {
			/* groups of four stripes, lane l takes dword l of each 64-byte group (xxh32_row) */
			const uint8_t *qr = p + 4 * l;
#define XXH_ROW_CHAIN(xg_)                                                                                        \
			do {                                                                                      \
				const uint32_t pr_ = (xg_) * XXH_P2;                                              \
				v = xxh_chain_step(v, pr_);                                                       \
				v = xxh_chain_step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pr_, 0x104, 0xf, 0xf, false)); \
				v = xxh_chain_step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pr_, 0x108, 0xf, 0xf, false)); \
				v = xxh_chain_step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pr_, 0x10c, 0xf, 0xf, false)); \
			} while (0)
			if (left >= XXH_ROW_BATCH * 64) {
				uint32_t x[XXH_ROW_BATCH];
#pragma unroll
				for (int g = 0; g < XXH_ROW_BATCH; g++)
					x[g] = ld_u32(qr + 64 * g);
				qr += XXH_ROW_BATCH * 64; left -= XXH_ROW_BATCH * 64;
				while (left >= XXH_ROW_BATCH * 64) {
					uint32_t y[XXH_ROW_BATCH];
#pragma unroll
					for (int g = 0; g < XXH_ROW_BATCH; g++)
						y[g] = ld_u32(qr + 64 * g);
#pragma unroll
					for (int g = 0; g < XXH_ROW_BATCH; g++)
						XXH_ROW_CHAIN(x[g]);
#pragma unroll
					for (int g = 0; g < XXH_ROW_BATCH; g++)
						x[g] = y[g];
					qr += XXH_ROW_BATCH * 64; left -= XXH_ROW_BATCH * 64;
				}
#pragma unroll
				for (int g = 0; g < XXH_ROW_BATCH; g++)
					XXH_ROW_CHAIN(x[g]);
			}
			while (left >= 64) {
				XXH_ROW_CHAIN(ld_u32(qr));
				qr += 64; left -= 64;
			}
#undef XXH_ROW_CHAIN
			q = p + (n - left) + 4 * j;	/* the last (fewer than four) stripes: lanes 0..3 on their own */
		}
		while (!ROW && left >= 512) {	/* 32 stripes of loads in flight per quad, as in xxh32_quad */
			uint32_t x[32];
#pragma unroll
			for (int t = 0; t < 32; t++)
				x[t] = ld_u32(q + 16 * t);
#pragma unroll
			for (int t = 0; t < 32; t++)
				v = xxh_round(v, x[t]);
			q += 512; left -= 512;
		}
		while (left >= 16) {
			v = xxh_round(v, ld_u32(q));
			q += 16; left -= 16;
		}
		const uint8_t *tailp = p + (n - left);
		for (uint32_t k = 0; k < (uint32_t)left; k++) mem[k] = tailp[k];
		msz = (uint32_t)left;
	}
	if (cout) {
		if (l < 4)	/* (in the row form only lanes 0..3 hold accumulators) */
			cout->v[j] = v;
		if (l == 0) {
			cout->memsize = msz;
			cout->total_lo = (uint32_t)total;
			cout->total_hi = (uint32_t)(total >> 32);
			for (int k = 0; k < 16; k++) cout->mem[k] = mem[k];
		}
		return 0;
	}
	/* digest (xxhash.c:447-507) */
	uint32_t h;
	const int rot = (j == 0) ? 1 : (j == 1) ? 7 : (j == 2) ? 12 : 18;
	uint32_t t = rotl32(v, rot);
	t += __shfl_xor(t, 1, 64);
	t += __shfl_xor(t, 2, 64);
	h = total >= 16 ? t : XXH_P5;
	h += (uint32_t)total;
	uint32_t k = 0;
	for (; k + 4 <= msz; k += 4) {
		const uint32_t dw = (uint32_t)mem[k] | ((uint32_t)mem[k + 1] << 8) | ((uint32_t)mem[k + 2] << 16) | ((uint32_t)mem[k + 3] << 24);
		h = rotl32(h + dw * XXH_P3, 17) * XXH_P4;
	}
	for (; k < msz; k++)
		h = rotl32(h + (uint32_t)mem[k] * XXH_P5, 11) * XXH_P1;
	return xxh_avalanche(h);
}

/* ROW = false: four lanes per frame (xxh32_quad; bulk).  ROW = true: sixteen lanes per frame
 * (xxh32_row; half the latency per hash, a quarter of the hashes per instruction) for launches
 * whose duration nothing hides. */
template <bool ROW>
__global__ __launch_bounds__(64) void lz4_frame_sums_kernel(const uint8_t *__restrict__ src,
    const uint8_t *__restrict__ dst, const la_lz4_frame *__restrict__ frames, uint32_t n,
    const uint64_t *__restrict__ dst_off, uint64_t dst_cap, uint32_t *__restrict__ fstatus,
    uint32_t end_lo, uint32_t end_hi, const la_xxh_carry *carry_in, la_xxh_carry *carry_out)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t i = ROW ? t >> 4 : t >> 2;
	const int l = (int)(t & (ROW ? 15u : 3u));
	int j = l & 3;
	if (i >= n)
		return;		/* n is rounded so that whole quads / rows leave together */
	la_lz4_frame f = frames[i];
	/* this launch takes the frames whose LAST block index + 1 lies in [end_lo, end_hi]
	 * (the batch is expanded in slices; a frame is hashed once all its blocks exist) */
	const uint32_t fend = f.first_block + f.n_blocks;
	if (fend < end_lo || fend > end_hi)
		return;
	uint32_t st = LA_ST_OK;
	if (f.flags & LA_LZ4F_HEADER_SUM) {
		uint32_t h = xxh32_quad(src + f.desc_off, f.desc_len - 1, 0, j);
		if (((h >> 8) & 0xff) != src[f.desc_off + f.desc_len - 1])
			st = LA_ST_LZ4_BAD_HEADER_SUM;
	}
	if (st == LA_ST_OK && (f.flags & (LA_LZ4F_CONT | LA_LZ4F_OPEN))) {
		/* a frame larger than one batch: its content hash comes in and / or goes out as state */
		if ((f.flags & LA_LZ4F_HASHED) && carry_in && carry_out) {
			uint64_t a = dst_off[f.first_block], e = dst_off[f.first_block + f.n_blocks];
			if (e <= dst_cap) {
				const uint32_t h = xxh32_quad_stream<ROW>((f.flags & LA_LZ4F_CONT) ? carry_in : nullptr,
				    (f.flags & LA_LZ4F_OPEN) ? carry_out : nullptr, dst + a, e - a, l);
				if (!(f.flags & LA_LZ4F_OPEN) && (f.flags & LA_LZ4F_CONTENT_SUM) && h != f.content_sum)
					st = LA_ST_LZ4_BAD_CONTENT_SUM;
			}
		}
	} else if (st == LA_ST_OK && (f.flags & LA_LZ4F_CONTENT_SUM)) {
		uint64_t a = dst_off[f.first_block], e = dst_off[f.first_block + f.n_blocks];
		if (e <= dst_cap) {
			/* xxhash.c:234: the length is an unsigned int there */
			uint32_t h = ROW ? xxh32_row(dst + a, (uint32_t)(e - a), 0, l) : xxh32_quad(dst + a, (uint32_t)(e - a), 0, j);
			if (h != f.content_sum)
				st = LA_ST_LZ4_BAD_CONTENT_SUM;
		}
	}
	if (l == 0)
		fstatus[i] = st;
}

void la_launch_xxh32_many(hipStream_t s, const uint8_t *d_base, const la_hash_job *d_jobs,
    uint32_t n, uint32_t *d_out)
{
	if (n == 0) return;
	hipLaunchKernelGGL(xxh32_many_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_base, d_jobs, n, d_out);
}

void la_launch_lz4_block_sums(hipStream_t s, const uint8_t *d_src, const la_lz4_block *d_blocks,
    uint32_t n, uint32_t *d_status)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_block_sums_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_src, d_blocks, n, d_status);
}

void la_launch_lz4_frame_sums(hipStream_t s, const uint8_t *d_src, const uint8_t *d_dst,
    const la_lz4_frame *d_frames, uint32_t n_frames, const uint64_t *d_dst_off, uint64_t dst_cap,
    uint32_t *d_frame_status, uint32_t end_lo, uint32_t end_hi, const void *d_carry_in, void *d_carry_out,
    int row)
{
	if (n_frames == 0) return;
	if (row)
		hipLaunchKernelGGL(lz4_frame_sums_kernel<true>, dim3((n_frames + 3) / 4), dim3(64), 0, s,
		    d_src, d_dst, d_frames, n_frames, d_dst_off, dst_cap, d_frame_status, end_lo, end_hi,
		    (const la_xxh_carry *)d_carry_in, (la_xxh_carry *)d_carry_out);
	else
		hipLaunchKernelGGL(lz4_frame_sums_kernel<false>, dim3((n_frames + 15) / 16), dim3(64), 0, s,
		    d_src, d_dst, d_frames, n_frames, d_dst_off, dst_cap, d_frame_status, end_lo, end_hi,
		    (const la_xxh_carry *)d_carry_in, (la_xxh_carry *)d_carry_out);
}

/* ------------------------------------------------------------------ CRC32 */

#define CRC_POLY 0xEDB88320u

/* a(x)*b(x) mod P(x), reflected bit order (bit 31 = x^0) */
__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
#pragma unroll 8
	for (int i = 0; i < 32; i++) {
		r ^= b & (0u - (a >> 31));
		a <<= 1;
		b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1)));
	}
	return r;
}

/* x^(8*nbytes) mod P */
__device__ uint32_t gf2_xpow8(uint32_t nbytes)
{
	uint32_t result = 0x80000000u, base = 0x00800000u;
	while (nbytes) {
		if (nbytes & 1)
			result = gf2_mulmod(result, base);
		base = gf2_mulmod(base, base);
		nbytes >>= 1;
	}
	return result;
}

/* slicing-by-4 tables in LDS: T[k][b] = state reached from byte b followed by k zero bytes */
__device__ __forceinline__ void crc_build_tables(uint32_t *T /* [4][256] */)
{
	for (uint32_t b = threadIdx.x; b < 256; b += blockDim.x) {
		uint32_t c = b;
		for (int k = 0; k < 8; k++)
			c = (c >> 1) ^ (CRC_POLY & (0u - (c & 1)));
		T[b] = c;
	}
	__syncthreads();
	for (uint32_t b = threadIdx.x; b < 256; b += blockDim.x) {
		uint32_t c = T[b];
		for (int k = 1; k < 4; k++) {
			c = T[c & 0xff] ^ (c >> 8);
			T[k * 256 + b] = c;
		}
	}
	__syncthreads();
}

/* raw CRC state update (no pre/post inversion) over n bytes by one lane */
__device__ __forceinline__ uint32_t crc_run(const uint32_t *T, uint32_t state, const uint8_t *p, uint32_t n)
{
	while (n && ((uintptr_t)p & 3)) {
		state = T[(state ^ *p++) & 0xff] ^ (state >> 8);
		n--;
	}
	/* 64 bytes per trip, the four 16-byte loads issued together: a lane walks its OWN chunk (the lanes of a wave are a
	 * whole chunk apart), so every 64-byte line is touched by one lane only -- loaded 16 bytes at a time the line
	 * was gone from the caches before its next quarter was asked for (round 3, rocprofv3 FETCH_SIZE: 54 GB fetched
	 * for the 17 GB of C3, profiles/r03_traffic_gzip.json) */
	while (n >= 64) {
		const uint4 *q = (const uint4 *)__builtin_assume_aligned(p, 4);
		const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
		const uint32_t w[16] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w };
#pragma unroll
		for (int k = 0; k < 16; k++) {
			uint32_t x = state ^ w[k];
			state = T[768 + (x & 0xff)] ^ T[512 + ((x >> 8) & 0xff)] ^
			    T[256 + ((x >> 16) & 0xff)] ^ T[x >> 24];
		}
		p += 64; n -= 64;
	}
	while (n >= 16) {
		uint4 v = *(const uint4 *)__builtin_assume_aligned(p, 4);
		uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int k = 0; k < 4; k++) {
			uint32_t x = state ^ w[k];
			state = T[768 + (x & 0xff)] ^ T[512 + ((x >> 8) & 0xff)] ^
			    T[256 + ((x >> 16) & 0xff)] ^ T[x >> 24];
		}
		p += 16; n -= 16;
	}
	while (n >= 4) {
		uint32_t x = state ^ *(const uint32_t *)p;
		state = T[768 + (x & 0xff)] ^ T[512 + ((x >> 8) & 0xff)] ^
		    T[256 + ((x >> 16) & 0xff)] ^ T[x >> 24];
		p += 4; n -= 4;
	}
	while (n--)
		state = T[(state ^ *p++) & 0xff] ^ (state >> 8);
	return state;
}

/*
 * One wave per range.  The range is laid out right-aligned in 64 virtual
 * chunks of S bytes (S a multiple of 16); lane i owns chunk i.  The lane that
 * holds the first real byte starts from the real initial state (seed ^ ~0),
 * every later lane from state 0, earlier lanes hold state 0 and no bytes.
 * Because the update is linear over GF(2), the state after the whole range is
 *     sum_i  state_i * x^(8 * S * (63 - i))      (mod P)
 * which the shuffle tree evaluates in 6 steps.
 */
__device__ __forceinline__ uint32_t crc32_wave(const uint32_t *T, const uint8_t *p, uint32_t len,
    uint32_t seed, int lane)
{
	if (len == 0)
		return seed;
	uint32_t S = ((len + 63) / 64 + 15) & ~15u;
	uint64_t pad = 64ull * S - len;		/* virtual bytes in front of the range */
	uint32_t f = (uint32_t)(pad / S);	/* lane holding the first real byte */
	uint64_t vbeg = (uint64_t)lane * S, vend = vbeg + S;
	uint32_t state = 0;
	if ((uint32_t)lane >= f) {
		uint64_t lo = vbeg < pad ? pad : vbeg;
		uint32_t init = ((uint32_t)lane == f) ? (seed ^ 0xFFFFFFFFu) : 0u;
		state = crc_run(T, init, p + (lo - pad), (uint32_t)(vend - lo));
	}
	uint32_t g = gf2_xpow8(S);
#pragma unroll
	for (int k = 0; k < 6; k++) {
		uint32_t right = __shfl_down(state, 1 << k, 64);
		state = gf2_mulmod(state, g) ^ right;
		g = gf2_mulmod(g, g);
	}
	return state ^ 0xFFFFFFFFu;	/* valid in lane 0 */
}

__global__ __launch_bounds__(256) void crc32_many_kernel(const uint8_t *__restrict__ base,
    const la_hash_job *__restrict__ jobs, uint32_t n, uint32_t *__restrict__ out)
{
	__shared__ uint32_t T[4 * 256];
	crc_build_tables(T);
	int lane = threadIdx.x & 63;
	uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t i = wave; i < n; i += nwaves) {
		la_hash_job j = jobs[i];
		uint32_t c = crc32_wave(T, base + j.off, j.len, j.seed, lane);
		if (lane == 0)
			out[i] = c;
	}
}

void la_launch_crc32_many(hipStream_t s, const uint8_t *d_base, const la_hash_job *d_jobs,
    uint32_t n, uint32_t *d_out)
{
	if (n == 0) return;
	uint32_t blocks = (n + 3) / 4;
	if (blocks > 256 * 8) blocks = 256 * 8;
	hipLaunchKernelGGL(crc32_many_kernel, dim3(blocks), dim3(256), 0, s, d_base, d_jobs, n, d_out);
}

/* ------------------------------------------------------------------ gzip trailer */

/*
 * One wave per member: CRC32 of the produced bytes (always reported), then the
 * 8-byte trailer [crc32 LE][isize LE] that follows the deflate body.  The
 * reference consumes the trailer without looking at it (gzip.c:423); a short
 * trailer is its message-less ARCHIVE_FATAL (gzip.c:419-421).
 */
__global__ __launch_bounds__(256) void gz_verify_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    const la_gz_member *__restrict__ members, uint32_t n, const uint8_t *__restrict__ dst,
    la_gz_result *results, int verify)
{
	__shared__ uint32_t T[4 * 256];
	crc_build_tables(T);
	int lane = threadIdx.x & 63;
	uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t i = wave; i < n; i += nwaves) {
		la_gz_member m = members[i];
		la_gz_result r = results[i];
		if (r.status != LA_ST_OK)
			continue;
		uint32_t c = crc32_wave(T, dst + m.dst_off, r.out_len, 0, lane);
		c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
		uint32_t st = LA_ST_OK;
		uint64_t tr = m.src_off + r.consumed;
		if (verify == 2)
			;	/* raw deflate member (LA_GZ_OPT_RAW): there is no trailer to look at */
		else if ((uint64_t)r.consumed + 8 > m.src_len || tr + 8 > src_bytes)
			st = LA_ST_GZ_NO_TRAILER;
		else if (verify) {
			if (ld_u32(src + tr) != c)
				st = LA_ST_GZ_BAD_CRC;
			else if (ld_u32(src + tr + 4) != r.out_len)
				st = LA_ST_GZ_BAD_ISIZE;
		}
		if (lane == 0) {
			results[i].crc32 = c;
			results[i].status = st;
		}
	}
}

void la_launch_gz_verify(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_gz_member *d_members, uint32_t n, const uint8_t *d_dst, la_gz_result *d_results, int verify)
{
	if (n == 0) return;
	uint32_t blocks = (n + 3) / 4;
	if (blocks > 256 * 8) blocks = 256 * 8;
	hipLaunchKernelGGL(gz_verify_kernel, dim3(blocks), dim3(256), 0, s, d_src, src_bytes, d_members, n,
	    d_dst, d_results, verify);
}

__global__ void gz_summary_init(la_batch_summary *sm)
{
	sm->total_out = 0; sm->n_bad_units = 0; sm->n_bad_frames = 0;
	sm->first_bad_unit = 0xFFFFFFFFu; sm->first_bad_frame = 0xFFFFFFFFu;
	sm->first_zero_unit = 0xFFFFFFFFu; sm->reserved = 0;
}

__global__ __launch_bounds__(256) void gz_summary_kernel(const la_gz_result *__restrict__ results, uint32_t n,
    la_batch_summary *sm)
{
	uint32_t stride = gridDim.x * blockDim.x;
	uint32_t bad = 0, first_bad = 0xFFFFFFFFu;
	unsigned long long total = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		la_gz_result r = results[i];
		total += r.out_len;
		if (r.status != LA_ST_OK) {
			bad++;
			if (i < first_bad) first_bad = i;
		}
	}
	for (int d = 32; d >= 1; d >>= 1) {
		bad += __shfl_down(bad, d, 64);
		total += __shfl_down(total, d, 64);
		first_bad = min(first_bad, (uint32_t)__shfl_down(first_bad, d, 64));
	}
	if ((threadIdx.x & 63) == 0) {
		if (bad) atomicAdd(&sm->n_bad_units, bad);
		if (total) atomicAdd((unsigned long long *)&sm->total_out, total);
		if (first_bad != 0xFFFFFFFFu) atomicMin(&sm->first_bad_unit, first_bad);
	}
}

void la_launch_gz_summary(hipStream_t s, const la_gz_result *d_results, uint32_t n, la_batch_summary *d_summary)
{
	hipLaunchKernelGGL(gz_summary_init, dim3(1), dim3(1), 0, s, d_summary);
	uint32_t blocks = (n + 255) / 256;
	if (blocks == 0) blocks = 1;
	if (blocks > 1024) blocks = 1024;
	hipLaunchKernelGGL(gz_summary_kernel, dim3(blocks), dim3(256), 0, s, d_results, n, d_summary);
}
