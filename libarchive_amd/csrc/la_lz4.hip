/*
 * la_lz4.hip -- LZ4 block decode kernels (gfx950).
 *
 * Replaces liblz4's LZ4_decompress_safe / LZ4_decompress_safe_usingDict as
 * called by libarchive/archive_read_support_filter_lz4.c:557-591 and :711, for a
 * whole table of blocks per launch.  Byte/integer work, HBM-bound; no MFMA.
 *
 * Two phases, because a block's decoded length is not in the frame:
 *   1. measure  -- one LANE per block walks the token chain (serial pointer
 *      chase, 64 chains per wave), applies every accept/reject rule of the
 *      library's safe decoder and reports decoded length + sequence count.
 *      An exclusive scan of the lengths (la_scan.hip) then packs the blocks
 *      back to back in the decoded slab.
 *   2. expand   -- copies literals and matches into the slab.
 *        general kernel (this file): one WAVE per block (or per chain of
 *        dependent blocks); token chain kept wave-uniform in SGPRs and fed from
 *        a 512-byte register window of the compressed payload via v_readlane;
 *        literal and match copies are wave-wide (64 bytes per step), matches
 *        read the already written slab.  Handles every block size (64 KiB..4 MiB,
 *        8 MiB legacy), stored blocks and dependent blocks.
 *        fast kernel (la_lz4_fast.hip): 64 KiB independent blocks through an LDS
 *        output window.
 */
#include "la_dev.h"

#define LZ4_MFLIMIT 12
#define LZ4_LASTLIT 5

/* ------------------------------------------------------------------ parse */

/*
 * One LANE per block.  The token chain is a serial pointer chase, so the lane
 * keeps an 8-byte window of the payload in registers: in the common case one
 * (unaligned) 8-byte load per sequence feeds offset, match-length extension
 * and the next token.  64 independent chains per wave hide the load latency.
 *
 * Accept/reject rules = liblz4 1.9.3 safe decoder (see oracle/orc_lz4.c for
 * the derivation): exact input consumption, last-sequence rules relative to
 * the destination CAPACITY, match end margin, offset range, bounded length
 * extensions; offset 0 rejected.
 *
 * With EMIT the lane also writes the block's sequence table (8 bytes per
 * sequence) for the LDS-window expand kernel.
 */
struct byte_window {
	const uint8_t *s;	/* payload */
	uint64_t w;
	int base;		/* payload offset of byte 0 of w */
	int limit;		/* payload offsets >= limit lie outside the source image */
};

__device__ __forceinline__ uint32_t bw_get(byte_window &B, int p)
{
	uint32_t d = (uint32_t)(p - B.base);
	if (d >= 8) {
		if (p + 8 <= B.limit) {
			__builtin_memcpy(&B.w, B.s + p, 8);
		} else {
			/* tail of the image: never read past it */
			uint64_t v = 0;
			for (int k = 0; k < 8; k++)
				if (p + k < B.limit)
					v |= (uint64_t)B.s[p + k] << (8 * k);
			B.w = v;
		}
		B.base = p;
		d = 0;
	}
	return (uint32_t)(B.w >> (8 * d)) & 0xffu;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void lz4_parse_kernel(const uint8_t *__restrict__ src,
    uint64_t src_bytes, const la_lz4_block *__restrict__ blocks, uint32_t n,
    uint32_t *__restrict__ out_len, uint32_t *__restrict__ nseq_out, uint32_t *__restrict__ status,
    la_lz4_seq *__restrict__ table, const uint64_t *__restrict__ table_off, uint64_t table_cap)
{
	/* Table entries are staged in LDS (transposed: conflict free) and leave in groups of
	 * eight = one aligned 64-byte store burst per lane.  Single 8-byte stores from 262144
	 * lanes to as many different cache lines exceed what L2 can hold until a line is
	 * complete, and every partial line costs a read-modify-write in HBM. */
	__shared__ uint64_t stage[8][256];
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	la_lz4_block b = blocks[i];
	if (status[i] != LA_ST_OK) {	/* block checksum already failed */
		out_len[i] = 0;
		nseq_out[i] = 0;
		return;
	}
	if (b.flags & LA_LZ4B_STORED) {
		out_len[i] = b.src_len;
		nseq_out[i] = 0;
		return;
	}
	byte_window B;
	B.s = src + b.src_off;
	B.base = -64;
	B.w = 0;
	int64_t room = (int64_t)src_bytes - (int64_t)b.src_off;
	B.limit = room < 0 ? 0 : (room > 0x7fffffff ? 0x7fffffff : (int)room);
	const int iend = (int)b.src_len;
	const int oend = (int)b.dst_cap;
	const int dict = (b.flags & LA_LZ4B_DEPENDENT) ? 65536 : 0;	/* lz4.c:563-584: any offset reaches the zero-filled prefix */
	/* emit only into a slot that lies inside the workspace (tables come from the host) */
	const bool eligible = EMIT && la_lz4_fast_eligible(b);
	const bool emit = eligible && table_off[i + 1] <= table_cap;
	uint64_t *tab = emit ? (uint64_t *)(table + table_off[i]) : nullptr;	/* slot is 64-byte aligned */
#define EMIT_SEQ(lit_src_, lit_len_, dst_, off_)                                                   \
	do {                                                                                       \
		stage[nseq & 7][threadIdx.x] = (uint64_t)(uint16_t)(lit_src_) |                    \
		    ((uint64_t)(uint16_t)(lit_len_) << 16) | ((uint64_t)(uint16_t)(dst_) << 32) |  \
		    ((uint64_t)(uint16_t)(off_) << 48);                                            \
		if ((nseq & 7) == 7) {                                                             \
			uint4 *o4 = (uint4 *)(tab + (nseq - 7));                                   \
			for (int j_ = 0; j_ < 4; j_++) {                                           \
				uint64_t a_ = stage[2 * j_][threadIdx.x], b_ = stage[2 * j_ + 1][threadIdx.x]; \
				o4[j_] = make_uint4((uint32_t)a_, (uint32_t)(a_ >> 32), (uint32_t)b_, (uint32_t)(b_ >> 32)); \
			}                                                                          \
		}                                                                                  \
	} while (0)
	int ip = 0, op = 0;
	uint32_t nseq = 0;
	bool ok = iend > 0;

	/* Blocks whose payload ends at least 8 bytes before the end of the image (all but the
	 * last one or two of a stream) take the lean loop: one unchecked 8-byte load at the
	 * token, a second one at the offset only when the literal run pushes it out of the
	 * first, length extensions (rare) through the careful byte reader.  Same rules, same
	 * results as the careful loop below. */
	const bool lean = (int64_t)iend + 8 <= room;
	while (ok && lean) {
		uint64_t w;
		__builtin_memcpy(&w, B.s + ip, 8);
		const uint32_t token = (uint32_t)w & 0xffu;
		int length = (int)(token >> 4);
		int hdr = 1;			/* bytes of w consumed so far */
		ip++;
		if (length == 15) {
			if (ip >= iend - 15) { ok = false; break; }
			uint32_t x;
			do {
				x = bw_get(B, ip++);
				length += (int)x;
				if (ip >= iend - 15 || length > oend)	/* (the sum must not wrap: longer than the output fails below) */
					break;
			} while (x == 255);
			hdr = 9;		/* the offset is not in w any more */
		}
		if (op + length > oend - LZ4_MFLIMIT || ip + length > iend - (2 + 1 + LZ4_LASTLIT)) {
			if (ip + length != iend || op + length > oend)
				ok = false;
			else if (emit && length > 0) {
				EMIT_SEQ(ip, length, op, 0);
				nseq++;
			}
			op += length;
			break;
		}
		const int lit_src = ip, lit_len = length, lit_dst = op;
		ip += length;
		op += length;
		/* offset (2 bytes) + first match-length extension byte */
		uint32_t o3;
		if (hdr + length + 3 <= 8)
			o3 = (uint32_t)(w >> (8 * (hdr + length)));
		else {
			uint64_t w2;
			__builtin_memcpy(&w2, B.s + ip, 8);
			o3 = (uint32_t)w2;
		}
		const int offset = (int)(o3 & 0xffffu);
		ip += 2;
		length = (int)(token & 15);
		if (length == 15) {
			uint32_t x = (o3 >> 16) & 0xffu;	/* first extension byte came with the offset */
			ip++;
			length += (int)x;
			if (ip >= iend - LZ4_LASTLIT + 1) { ok = false; break; }
			while (x == 255) {
				x = bw_get(B, ip++);
				length += (int)x;
				if (ip >= iend - LZ4_LASTLIT + 1 || length > oend) { ok = false; break; }
			}
			if (!ok) break;
		}
		length += 4;
		if (offset == 0 || offset > op + dict || op + length > oend - LZ4_LASTLIT) { ok = false; break; }
		if (emit)
			EMIT_SEQ(lit_src, lit_len, lit_dst, offset);
		nseq++;
		op += length;
	}
	while (ok && !lean) {
		uint32_t token = bw_get(B, ip++);
		int length = (int)(token >> 4);
		if (length == 15) {
			if (ip >= iend - 15) { ok = false; break; }
			uint32_t x;
			do {
				x = bw_get(B, ip++);
				length += (int)x;
				if (ip >= iend - 15 || length > oend)	/* (the sum must not wrap: longer than the output fails below) */
					break;
			} while (x == 255);
		}
		if (op + length > oend - LZ4_MFLIMIT || ip + length > iend - (2 + 1 + LZ4_LASTLIT)) {
			if (ip + length != iend || op + length > oend)
				ok = false;
			else if (emit && length > 0) {
				EMIT_SEQ(ip, length, op, 0);
				nseq++;
			}
			op += length;
			break;
		}
		const int lit_src = ip, lit_len = length, lit_dst = op;
		ip += length;
		op += length;
		int offset = (int)bw_get(B, ip) | ((int)bw_get(B, ip + 1) << 8);
		ip += 2;
		length = (int)(token & 15);
		if (length == 15) {
			uint32_t x;
			do {
				x = bw_get(B, ip++);
				length += (int)x;
				if (ip >= iend - LZ4_LASTLIT + 1 || length > oend) { ok = false; break; }
			} while (x == 255);
			if (!ok) break;
		}
		length += 4;
		if (offset == 0 || offset > op + dict || op + length > oend - LZ4_LASTLIT) { ok = false; break; }
		if (emit)
			EMIT_SEQ(lit_src, lit_len, lit_dst, offset);
		nseq++;
		op += length;
	}
	if (emit && ok)
		for (uint32_t j = nseq & ~7u; j < nseq; j++)	/* the last, incomplete group of eight */
			tab[j] = stage[j & 7][threadIdx.x];
#undef EMIT_SEQ
	out_len[i] = ok ? (uint32_t)op : 0u;
	/* an eligible block without a table slot must go to the general kernel */
	nseq_out[i] = ok ? ((eligible && !emit) ? 0xFFFFFFFFu : nseq) : 0u;
	if (!ok)
		status[i] = LA_ST_LZ4_DECODE;
}

/* table capacity per block: a non-final sequence takes at least 3 payload bytes */
__global__ __launch_bounds__(256) void lz4_table_caps_kernel(const la_lz4_block *__restrict__ blocks,
    uint32_t n, uint32_t *__restrict__ caps)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	la_lz4_block b = blocks[i];
	caps[i] = la_lz4_fast_eligible(b) ? (b.src_len / 3 + 1 + 7) & ~7u : 0u;	/* slots stay 64-byte aligned */
}

void la_launch_lz4_table_caps(hipStream_t s, const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_caps)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_table_caps_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_blocks, n, d_caps);
}

void la_launch_lz4_parse(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_out_len, uint32_t *d_nseq,
    uint32_t *d_status, la_lz4_seq *d_table, const uint64_t *d_table_off, uint64_t table_cap)
{
	if (n == 0) return;
	if (d_table)
		hipLaunchKernelGGL(lz4_parse_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, s,
		    d_src, src_bytes, d_blocks, n, d_out_len, d_nseq, d_status, d_table, d_table_off, table_cap);
	else
		hipLaunchKernelGGL(lz4_parse_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, s,
		    d_src, src_bytes, d_blocks, n, d_out_len, d_nseq, d_status, d_table, d_table_off, table_cap);
}

/* ------------------------------------------------------------------ general expand */

/*
 * Decode one validated block with one wave.  d points at the block's slot in
 * the slab; bytes d[-dict_len..0) are the previous block (dependent frames),
 * anything further back reads as zero (the reference keeps only the previous
 * block in its 64 KiB prefix and zero-fills the rest, lz4.c:563-577).
 */
__device__ void lz4_expand_block_wave(const uint8_t *s, uint32_t src_len, const uint8_t *src_limit,
    uint8_t *d, uint32_t dict_len, int lane)
{
	src_window W;
	W.limit = src_limit;
	win_reset(W, s, lane);
	const uint8_t *ip = s;
	const uint8_t *iend = s + src_len;
	uint32_t op = 0;
	uint32_t visible = 0;	/* output bytes [0, visible) are known to be visible to loads */

	for (;;) {
		uint32_t token = win_byte(W, ip++, lane);
		uint32_t lit = token >> 4;
		if (lit == 15) {
			uint32_t x;
			do {
				x = win_byte(W, ip++, lane);
				lit += x;
			} while (x == 255);
		}
		/* literals: 64 bytes per step, coalesced */
		for (uint32_t j = (uint32_t)lane; j < lit; j += LA_WAVE)
			d[op + j] = ip[j];
		ip += lit;
		op += lit;
		if (ip >= iend)
			break;
		uint32_t off = win_byte(W, ip, lane) | (win_byte(W, ip + 1, lane) << 8);
		ip += 2;
		uint32_t ml = token & 15;
		if (ml == 15) {
			uint32_t x;
			do {
				x = win_byte(W, ip++, lane);
				ml += x;
			} while (x == 255);
		}
		ml += 4;

		/* Source bytes are [op-off, op-off+min(ml,off)); with the modulo form
		 * every source byte lies below op, i.e. was stored by an EARLIER
		 * instruction of this wave.  Fence only if some of it is newer than
		 * the last fence. */
		uint32_t span = ml < off ? ml : off;
		if (op > visible && (int64_t)op - (int64_t)off + (int64_t)span > (int64_t)visible) {
			wave_mem_fence();
			visible = op;
		}
		for (uint32_t j = (uint32_t)lane; j < ml; j += LA_WAVE) {
			uint32_t k = (off < ml) ? (j % off) : j;
			int64_t from = (int64_t)op - (int64_t)off + (int64_t)k;
			uint8_t v;
			if (from >= 0 || (uint64_t)(-from) <= dict_len)
				v = d[from];
			else
				v = 0;
			d[op + j] = v;
		}
		op += ml;
	}
}

__global__ __launch_bounds__(256) void lz4_expand_general_kernel(const uint8_t *__restrict__ src,
    uint64_t src_bytes, const la_lz4_block *__restrict__ blocks, uint32_t n, uint8_t *dst,
    uint64_t dst_cap, const uint64_t *__restrict__ dst_off, const uint32_t *__restrict__ out_len,
    const uint32_t *__restrict__ status, const uint32_t *__restrict__ nseq, uint32_t fast_max_seq, uint32_t hist_len,
    uint32_t long_thr)
{
	int lane = threadIdx.x & 63;
	uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	if (i >= n)
		return;
	uint32_t fl = blocks[i].flags;
	/* a dependent block that is not the first of its frame belongs to the
	 * chain of the wave that owns the frame's first block */
	if ((fl & LA_LZ4B_DEPENDENT) && !(fl & (LA_LZ4B_FIRST | LA_LZ4B_HIST)))
		return;
	const uint8_t *src_limit = src + src_bytes;
	/* a chain that continues a frame from the previous batch starts with that batch's last
	 * block as its dictionary: the caller has put those bytes in front of the slab */
	uint32_t prev_len = (fl & LA_LZ4B_HIST) ? hist_len : 0;
	for (;;) {
		la_lz4_block b = blocks[i];
		uint8_t *d = dst + dst_off[i];
		uint32_t olen = out_len[i];
		/* never write past the slab, whatever the tables say; blocks the LDS-window
		 * kernel takes are skipped here (same predicate on both sides) */
		if (status[i] == LA_ST_OK && olen > 0 && dst_off[i] + olen <= dst_cap &&
		    !(fast_max_seq && la_lz4_fast_eligible(b) && nseq[i] <= fast_max_seq && !la_lz4_long_sequences(nseq[i], olen, long_thr))) {
			const uint8_t *s = src + b.src_off;
			if (b.flags & LA_LZ4B_STORED) {
				/* stored block (lz4.c:530-552): bytes up to the slab's 16-byte grid one by one,
				 * then 1 KiB per wave and pass (unaligned 16-byte loads, aligned stores) */
				uint32_t head = (16u - (uint32_t)((uintptr_t)d & 15u)) & 15u;
				if (head > b.src_len) head = b.src_len;
				if ((uint32_t)lane < head)
					d[lane] = s[lane];
				const uint32_t nvec = (b.src_len - head) >> 4;
				for (uint32_t j = (uint32_t)lane; j < nvec; j += LA_WAVE)
					*(uint4 *)(d + head + 16u * j) = ld_u128(s + head + 16u * j);
				for (uint32_t j = head + (nvec << 4) + (uint32_t)lane; j < b.src_len; j += LA_WAVE)
					d[j] = s[j];
			} else {
				uint32_t dict_len = 0;
				if ((b.flags & LA_LZ4B_DEPENDENT) && !(b.flags & LA_LZ4B_FIRST))
					dict_len = prev_len < 65536u ? prev_len : 65536u;
				lz4_expand_block_wave(s, b.src_len, src_limit, d, dict_len, lane);
			}
		}
		if (!(b.flags & LA_LZ4B_DEPENDENT))
			break;
		/* next block of the chain */
		prev_len = olen;
		i++;
		if (i >= n)
			break;
		uint32_t nf = blocks[i].flags;
		if (!(nf & LA_LZ4B_DEPENDENT) || (nf & (LA_LZ4B_FIRST | LA_LZ4B_HIST)))
			break;
		wave_mem_fence();	/* previous block's bytes become the dictionary */
	}
}

void la_launch_lz4_expand_general(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint8_t *d_dst, uint64_t dst_cap,
    const uint64_t *d_dst_off, const uint32_t *d_out_len, const uint32_t *d_status,
    const uint32_t *d_nseq, uint32_t fast_max_seq, uint32_t hist_len, uint32_t long_thr)
{
	if (n == 0) return;
	hipLaunchKernelGGL(lz4_expand_general_kernel, dim3((n + 3) / 4), dim3(256), 0, s,
	    d_src, src_bytes, d_blocks, n, d_dst, dst_cap, d_dst_off, d_out_len, d_status, d_nseq, fast_max_seq, hist_len, long_thr);
}
