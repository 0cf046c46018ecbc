/*
 * la_deflate_comp.hip -- DEFLATE compression + gzip member assembly on the device (gfx950):
 * the data plane of the gzip write filter.
 *
 * Replaces, for a whole stream per call, what libarchive/archive_write_add_filter_gzip.c does
 * through zlib on the host: deflateInit2(-15) / deflate() per write (:293-345, drive_compressor),
 * the 10-byte header it builds by hand (:201-237), CRC32 of the input (:263-266) and the trailer
 * [crc32 LE][isize LE] (:309-331).  The compressed bytes are not zlib's (a deflate stream is not
 * unique); parity for this direction is the round trip: zlib's inflate (what every gzip reader runs,
 * the reference's included), the oracle's gzip filter and this repository's device decoder must
 * return the input.
 *
 * Shape.  The input is cut into chunks of at most 48 KiB and every chunk becomes ONE gzip member --
 * the many-member shape the read side is built for -- whose header carries the BGZF-compatible "BC"
 * size subfield, so that the read filter indexes members without searching (C3's stream shape).
 *   deflate_fixed_kernel    one wave per chunk.  Per window of 64 positions every lane hashes three bytes,
 *       takes and replaces the candidate in a 4096-entry table of 16-bit positions (LDS), verifies it
 *       and extends the match to at most 258 bytes within 32 KiB; the wave takes the matches in position
 *       order and marks what they cover; then every lane knows its token -- literal, match start or
 *       nothing -- as at most 31 bits of a FIXED-Huffman block (RFC 1951 3.2.6), a wave prefix sum of
 *       the bit counts gives every token its place, lanes OR their bits into a small LDS stage and whole
 *       dwords go out coalesced.  A chunk that would not shrink is written as a stored block instead.
 *   gz_jobs_kernel + crc32_many  CRC32 of every chunk (la_hash.hip).
 *   gz_pack_members_kernel  header, body (fixed-Huffman or stored), trailer at their scanned offsets.
 */
#include "la_dev.h"

#define DFL_CHUNK_MAX 49152u
#define DFL_HASH_BITS 12
#define DFL_HDR       18u	/* 10 fixed + XLEN(2) + "BC" 2 0 BSIZE(2) */

__host__ __device__ static inline uint32_t dfl_body_bound(uint32_t n) { return ((n * 9u + 7u) >> 3) + 16u; }

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t n) { return __builtin_bitreverse32(v) >> (32u - n); }

/* literal / end-of-block code of the fixed tree, ready for LSB-first packing; *nb = its length */
__device__ __forceinline__ uint32_t fixed_lit(uint32_t sym, uint32_t *nb)
{
	if (sym < 144u) { *nb = 8; return rev_bits(0x30u + sym, 8); }
	if (sym < 256u) { *nb = 9; return rev_bits(0x190u + (sym - 144u), 9); }
	if (sym < 280u) { *nb = 7; return rev_bits(sym - 256u, 7); }
	*nb = 8; return rev_bits(0xC0u + (sym - 280u), 8);
}

/* a match of `len` (3..258) at `dist` (1..32768) as bits; returns the bit count (at most 31) */
__device__ __forceinline__ uint32_t fixed_match(uint32_t len, uint32_t dist, uint32_t *bits)
{
	uint32_t l = len - 3u, lcode, leb, lex;
	if (l < 8u) { lcode = 257u + l; leb = 0; lex = 0; }
	else if (l == 255u) { lcode = 285u; leb = 0; lex = 0; }
	else {
		const uint32_t n = 31u - (uint32_t)__builtin_clz(l);
		leb = n - 2u;
		lcode = 257u + 4u * (n - 1u) + ((l - (1u << n)) >> leb);
		lex = (l - (1u << n)) & ((1u << leb) - 1u);
	}
	uint32_t d = dist - 1u, dcode, deb, dex;
	if (d < 4u) { dcode = d; deb = 0; dex = 0; }
	else {
		const uint32_t n = 31u - (uint32_t)__builtin_clz(d);
		deb = n - 1u;
		dcode = 2u * n + ((d >> (n - 1u)) & 1u);
		dex = d & ((1u << deb) - 1u);
	}
	uint32_t nb, v = fixed_lit(lcode, &nb);
	v |= lex << nb; nb += leb;
	v |= rev_bits(dcode, 5) << nb; nb += 5u;
	v |= dex << nb; nb += deb;
	*bits = v;
	return nb;
}

__global__ __launch_bounds__(64) void deflate_fixed_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    uint32_t chunk, uint32_t n_chunks, uint8_t *__restrict__ tmp, uint32_t tmp_stride, uint32_t *__restrict__ body_len)
{
	__shared__ uint16_t tab[1u << DFL_HASH_BITS];
	__shared__ uint32_t stage[72];
	const uint32_t ci = blockIdx.x, lane = threadIdx.x;
	if (ci >= n_chunks)
		return;
	const uint64_t so = (uint64_t)ci * chunk;
	const uint32_t n = (uint32_t)(src_bytes - so < chunk ? src_bytes - so : chunk);
	const uint8_t *in = src + so;
	uint32_t *out = (uint32_t *)(void *)(tmp + (uint64_t)ci * tmp_stride);	/* dword aligned: the stride is a multiple of 16 */
	for (uint32_t i = lane; i < (1u << DFL_HASH_BITS); i += 64)
		tab[i] = 0;
	for (uint32_t i = lane; i < 72; i += 64)
		stage[i] = 0;
	__syncthreads();
	if (lane == 0)
		stage[0] = 3u;	/* block header: BFINAL = 1, BTYPE = 01 (fixed Huffman), LSB first */
	uint64_t bp = 3;	/* bits written so far (wave-uniform) */
	uint32_t anchor = 0;	/* first position not covered by a match taken so far */
	__builtin_amdgcn_wave_barrier();

	for (uint32_t base = 0; base < n; base += 64) {
		const uint32_t p = base + lane;
		const bool have = p < n;
		uint32_t cand = 0, mlen = 0, v3 = 0;
		const bool can = have && p + 3u <= n;
		if (can) {
			v3 = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
			cand = tab[(v3 * 2654435761u) >> (32 - DFL_HASH_BITS)];
		}
		__builtin_amdgcn_wave_barrier();
		bool ok = false;
		if (can) {
			tab[(v3 * 2654435761u) >> (32 - DFL_HASH_BITS)] = (uint16_t)p;
			if (cand < p && p - cand <= 32768u) {
				const uint32_t c3 = (uint32_t)in[cand] | ((uint32_t)in[cand + 1] << 8) | ((uint32_t)in[cand + 2] << 16);
				if (c3 == v3) {
					const uint32_t lim = n - p < 258u ? n - p : 258u;
					mlen = 3;
					while (mlen < lim && in[p + mlen] == in[cand + mlen])
						mlen++;
					ok = true;
				}
			}
		}
		/* the matches of this window in position order; `covered` = inside a match taken earlier */
		bool covered = have && p < anchor, taken = false;
		uint64_t mask = __ballot(ok);
		while (mask != 0) {
			const uint32_t f = (uint32_t)__builtin_ctzll(mask);
			mask &= mask - 1;
			const uint32_t pf = base + f;
			if (pf < anchor)
				continue;
			const uint32_t mf = (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)f);
			if (lane == f)
				taken = true;
			covered = covered || (p > pf && p < pf + mf);
			anchor = pf + mf;
		}
		/* this lane's token */
		uint32_t bits = 0, nb = 0;
		if (have && !covered) {
			if (taken)
				nb = fixed_match(mlen, p - cand, &bits);
			else
				bits = fixed_lit(in[p], &nb);
		}
		/* its place: exclusive prefix sum of the bit counts over the wave */
		uint32_t inc = nb;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(inc, d, 64);
			if ((int)lane >= d) inc += t;
		}
		const uint32_t total = __shfl(inc, 63, 64);
		const uint32_t at = (uint32_t)(bp & 31u) + inc - nb;	/* bit offset inside the stage */
		if (nb) {
			const uint64_t w = (uint64_t)bits << (at & 31u);
			atomicOr(&stage[at >> 5], (uint32_t)w);
			if ((uint32_t)(w >> 32))
				atomicOr(&stage[(at >> 5) + 1], (uint32_t)(w >> 32));
		}
		__builtin_amdgcn_wave_barrier();
		/* whole dwords leave; the partial last one stays as the next window's first */
		const uint32_t tb = (uint32_t)(bp & 31u) + total, nd = tb >> 5;
		const uint32_t g0 = (uint32_t)(bp >> 5);
		uint32_t mine = 0, carry = stage[nd];
		if (lane < nd)
			mine = stage[lane];
		__builtin_amdgcn_wave_barrier();
		if (lane < nd)
			out[g0 + lane] = mine;
		if (lane <= nd && lane < 72)
			stage[lane] = 0;
		if (nd >= 64 && lane == 0) {	/* (at most 64 * 31 + 31 bits: dword 64 can only be the partial one) */
			stage[64] = 0;
		}
		__builtin_amdgcn_wave_barrier();
		if (lane == 0)
			stage[0] = carry;
		__builtin_amdgcn_wave_barrier();
		bp += total;
	}
	/* end-of-block (seven zero bits), then the partial dword */
	bp += 7;
	{
		const uint32_t g0 = (uint32_t)((bp - 7) >> 5);
		const uint32_t tb = (uint32_t)((bp - 7) & 31u) + 7u;
		if (lane == 0) {
			out[g0] = stage[0];
			if (tb > 32u)
				out[g0 + 1] = 0;
		}
	}
	if (lane == 0)
		body_len[ci] = (uint32_t)((bp + 7) >> 3);
}

__global__ __launch_bounds__(256) void gz_jobs_kernel(uint64_t src_bytes, uint32_t chunk, uint32_t n_chunks,
    const uint32_t *__restrict__ body_len, la_hash_job *__restrict__ jobs, uint32_t *__restrict__ contrib)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_chunks)
		return;
	const uint64_t so = (uint64_t)i * chunk;
	const uint32_t n = (uint32_t)(src_bytes - so < chunk ? src_bytes - so : chunk);
	jobs[i].off = so; jobs[i].len = n; jobs[i].seed = 0;
	const uint32_t body = body_len[i] < n + 5u ? body_len[i] : n + 5u;	/* stored block: 01 LEN NLEN data */
	contrib[i] = DFL_HDR + body + 8u;
}

__global__ __launch_bounds__(256) void gz_pack_members_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    uint32_t chunk, uint32_t n_chunks, uint32_t mtime, const uint8_t *__restrict__ tmp, uint32_t tmp_stride,
    const uint32_t *__restrict__ body_len, const uint32_t *__restrict__ crc, const uint64_t *__restrict__ off,
    uint8_t *__restrict__ out, uint64_t out_cap, uint64_t *__restrict__ out_bytes)
{
	const uint32_t ci = blockIdx.x, tid = threadIdx.x;
	if (ci >= n_chunks)
		return;
	const uint64_t so = (uint64_t)ci * chunk;
	const uint32_t n = (uint32_t)(src_bytes - so < chunk ? src_bytes - so : chunk);
	const bool stored = body_len[ci] >= n + 5u;
	const uint32_t body = stored ? n + 5u : body_len[ci];
	const uint64_t o = off[ci];
	if (ci + 1 == n_chunks && tid == 0)
		*out_bytes = off[n_chunks];
	if (off[ci + 1] > out_cap)
		return;
	const uint32_t total = DFL_HDR + body + 8u;
	if (tid == 0) {
		uint8_t *h = out + o;
		h[0] = 0x1f; h[1] = 0x8b; h[2] = 8; h[3] = 4;	/* FEXTRA */
		h[4] = (uint8_t)mtime; h[5] = (uint8_t)(mtime >> 8); h[6] = (uint8_t)(mtime >> 16); h[7] = (uint8_t)(mtime >> 24);
		h[8] = 0; h[9] = 3;				/* XFL 0, OS = Unix (archive_write_add_filter_gzip.c:230-231) */
		h[10] = 6; h[11] = 0; h[12] = 'B'; h[13] = 'C'; h[14] = 2; h[15] = 0;
		h[16] = (uint8_t)(total - 1u); h[17] = (uint8_t)((total - 1u) >> 8);
		uint8_t *t = out + o + DFL_HDR + body;
		const uint32_t c = crc[ci];
		t[0] = (uint8_t)c; t[1] = (uint8_t)(c >> 8); t[2] = (uint8_t)(c >> 16); t[3] = (uint8_t)(c >> 24);
		t[4] = (uint8_t)n; t[5] = (uint8_t)(n >> 8); t[6] = (uint8_t)(n >> 16); t[7] = (uint8_t)(n >> 24);
	}
	uint8_t *b = out + o + DFL_HDR;
	if (stored) {
		if (tid == 0) {
			b[0] = 1;	/* BFINAL = 1, BTYPE = 00 */
			b[1] = (uint8_t)n; b[2] = (uint8_t)(n >> 8); b[3] = (uint8_t)~n; b[4] = (uint8_t)(~n >> 8);
		}
		for (uint32_t i = tid; i < n; i += 256)
			b[5 + i] = src[so + i];
	} else {
		const uint8_t *t = tmp + (uint64_t)ci * tmp_stride;
		for (uint32_t i = tid; i < body; i += 256)
			b[i] = t[i];
	}
}

extern "C" uint64_t la_gpu_gzip_compress_workspace_bytes(uint64_t src_bytes, uint32_t chunk)
{
	if (chunk == 0)
		return 0;
	const uint64_t nc = (src_bytes + chunk - 1) / chunk;
	const uint64_t stride = (dfl_body_bound(chunk) + 15u) & ~15ull;
	return nc * stride + nc * (4 + 4 + 4 + sizeof(la_hash_job)) + (nc + 1) * 8 + la_scan_scratch_bytes((uint32_t)nc) + 4096;
}

void la_launch_gzip_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes, uint32_t chunk, uint32_t mtime,
    uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_bytes, uint8_t *ws)
{
	const uint32_t nc = (uint32_t)((src_bytes + chunk - 1) / chunk);
	const uint32_t stride = (dfl_body_bound(chunk) + 15u) & ~15u;
	uint64_t o = 0;
	uint8_t *tmp = ws + o; o += (uint64_t)nc * stride;
	uint32_t *body_len = (uint32_t *)(ws + o); o += (uint64_t)nc * 4;
	uint32_t *contrib = (uint32_t *)(ws + o); o += (uint64_t)nc * 4;
	uint32_t *crc = (uint32_t *)(ws + o); o += (uint64_t)nc * 4;
	o = (o + 15) & ~15ull;
	la_hash_job *jobs = (la_hash_job *)(ws + o); o += (uint64_t)nc * sizeof(la_hash_job);
	uint64_t *off = (uint64_t *)(ws + o); o += ((uint64_t)nc + 1) * 8;
	o = (o + 255) & ~255ull;
	void *scan = ws + o;
	if (nc == 0) {
		(void)hipMemsetAsync(d_out_bytes, 0, 8, s);
		return;
	}
	hipLaunchKernelGGL(deflate_fixed_kernel, dim3(nc), dim3(64), 0, s, d_src, src_bytes, chunk, nc, tmp, stride, body_len);
	hipLaunchKernelGGL(gz_jobs_kernel, dim3((nc + 255) / 256), dim3(256), 0, s, src_bytes, chunk, nc, body_len, jobs, contrib);
	la_launch_crc32_many(s, d_src, jobs, nc, crc);
	la_launch_scan_u32(s, contrib, nc, off, scan);
	hipLaunchKernelGGL(gz_pack_members_kernel, dim3(nc), dim3(256), 0, s, d_src, src_bytes, chunk, nc, mtime, tmp, stride,
	    body_len, crc, off, d_out, out_cap, d_out_bytes);
}
