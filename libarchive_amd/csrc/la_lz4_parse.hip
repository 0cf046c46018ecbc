/*
 * la_lz4_parse.hip -- LZ4 token-chain parse + block checksums, one pass over the
 * compressed image (gfx950).
 *
 * Replaces, for a whole table of blocks per launch, the two things
 * libarchive/archive_read_support_filter_lz4.c does with a block's compressed bytes
 * before any output exists: the block checksum (lz4.c:507-524, __archive_xxhash
 * XXH32 over the payload) and the library decoder's walk of the token chain
 * (LZ4_decompress_safe, called at lz4.c:557-591), here reduced to its accept/reject
 * rules, the decoded length and the sequence table the expand kernels consume.
 *
 * One LANE per block, as in lz4_parse_kernel (la_lz4.hip), but the lanes no longer
 * touch global memory for payload bytes.  A lane-per-block walk reads 64 different
 * cache lines per wave instruction, a few bytes of each, and by the time a lane comes
 * back for the rest of its line it has left the cache: the v1 kernel fetched 3.3x the
 * image from HBM and spent 9 ms on 7.5 GB.  Here the wave stages the image through
 * LDS in rounds:
 *
 *   round r   the wave owns chunks r and r+1 (CH bytes each, cut on the IMAGE's CH-byte
 *             grid starting at the grid point before the block) of each of its 64
 *             blocks in an LDS ring (dword-transposed: ring[dword][lane], so a lane's
 *             reads never conflict with another lane's).  Chunk r+2 is in flight in
 *             registers.
 *   load      CH/16 wave instructions per round; in each, CH/16 neighbouring lanes
 *             read one block's chunk as consecutive 16-byte pieces = one whole,
 *             aligned cache line, fetched exactly once.
 *   parse     every lane walks its own token chain out of the ring for as long as
 *             its token is staged (it may run up to one chunk ahead of the round,
 *             which evens out lanes whose sequences are short or long).
 *   checksum  XXH32 stripes of chunk r, four accumulators per lane, from the same
 *             staged bytes: the image is read from HBM exactly once for both jobs.
 *
 * (Tried without effect: branch-free chunk loads -- as written the compiler waits for each
 * 16-byte load right behind it, because the byte-wise image-tail path defines the same
 * registers; with all eight loads of a round in flight the kernel was not faster, 5.6 vs 5.2 ms.)
 *
 * Accept/reject rules are those of lz4_parse_kernel (= liblz4 1.9.3's safe decoder,
 * derivation in oracle/orc_lz4.c); tests run both kernels on the same tables and
 * require identical out_len / nseq / status / table bytes.
 */
#include "la_dev.h"

#define LZ4_MFLIMIT 12
#define LZ4_LASTLIT 5

#ifndef PS_CH
#define PS_CH 128u		/* bytes of every block staged per round (multiple of 16).  Measured on the
				 * 16 GiB workload: 32 -> 9.9 ms, 64 -> 6.0, 128 -> 5.2, 256 -> 8.4 (a larger
				 * ring lets lanes drift further apart before they wait for each other, but
				 * LDS per wave sets how many waves a CU holds) */
#endif
#ifndef PS_THREADS
#define PS_THREADS 64u		/* waves of a workgroup share nothing: the size only sets the LDS granule */
#endif
#ifndef PS_STAGE
#define PS_STAGE   16u		/* staged table entries per lane (two groups of eight) */
#endif
#define PS_WAVES   (PS_THREADS / 64u)
#define PS_RINGW   (2u * PS_CH / 4u)	/* ring = two chunks, in dwords (power of two) */
#define PS_PIECES  (PS_CH / 16u)	/* 16-byte pieces per chunk = load instructions per round */
#define PS_BPI     (64u / PS_PIECES)	/* blocks covered by one load instruction */

struct ps_loader {
	uint64_t ptr[PS_PIECES];	/* image offset of this lane's piece 0-chunk for instruction i */
	uint32_t need[PS_PIECES];	/* payload bytes to stage for that block */
	uint64_t room[PS_PIECES];	/* bytes of the image from the block's start on */
};

/* issue the loads of chunk c (PS_PIECES x 16 bytes per lane) */
__device__ __forceinline__ void ps_load_chunk(const uint8_t *__restrict__ src, const ps_loader &L,
    uint32_t c, uint32_t piece, uint4 (&fl)[PS_PIECES])
{
#pragma unroll
	for (uint32_t i = 0; i < PS_PIECES; i++) {
		const uint64_t co = (uint64_t)c * PS_CH + 16u * piece;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (co < L.need[i]) {
			const uint8_t *p = src + L.ptr[i] + co;
			if (co + 16 <= L.room[i]) {
				v = ld_u128(p);
			} else {
				/* last piece of the image: never read past it */
				uint32_t d[4] = { 0, 0, 0, 0 };
				for (uint32_t k = 0; k < 16 && co + k < L.room[i]; k++)
					d[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
				v = make_uint4(d[0], d[1], d[2], d[3]);
			}
		}
		fl[i] = v;
	}
}

template <bool EMIT, bool SUMS>
__global__ __launch_bounds__(PS_THREADS) void lz4_parse_staged_kernel(const uint8_t *__restrict__ src,
    uint64_t src_bytes, const la_lz4_block *__restrict__ blocks, uint32_t n,
    uint32_t *__restrict__ out_len, uint32_t *__restrict__ nseq_out, uint32_t *__restrict__ status,
    uint32_t *__restrict__ sum_status, la_lz4_seq *__restrict__ table,
    const uint64_t *__restrict__ table_off, uint64_t table_cap)
{
	__shared__ uint32_t ring[PS_WAVES][PS_RINGW + 4u][64];	/* last rows = copies of rows 0..3: up to five consecutive dwords never wrap */
	/* table entries wait here (transposed: conflict free) and leave in groups of eight = one
	 * aligned 64-byte burst per lane (see la_lz4.hip), at the START of a round: the stores
	 * then have a whole round of LDS-only work to complete in before the wave next waits on
	 * its memory counter (gfx950 counts loads and stores in one in-order counter, so a store
	 * issued just before the wait for the next chunk would put its full latency on the path) */
	__shared__ uint64_t stage[PS_STAGE][PS_THREADS];

	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t i = blockIdx.x * PS_THREADS + threadIdx.x;
	const bool have = i < n;
	uint32_t (*R)[64] = ring[wave];

	la_lz4_block b;
	b.src_off = 0; b.src_len = 0; b.dst_cap = 0; b.flags = LA_LZ4B_STORED; b.block_sum = 0;
	if (have)
		b = blocks[i];
	const bool pre_failed = have && status[i] != LA_ST_OK;
	const bool do_parse = have && !pre_failed && !(b.flags & LA_LZ4B_STORED);	/* (a payload of 2^31 bytes or more fails below: iend <= 0) */
	const bool do_sum = SUMS && have && (b.flags & LA_LZ4B_CHECKSUM);
	const uint64_t room64 = b.src_off < src_bytes ? src_bytes - b.src_off : 0;
	/* Chunks are cut on the IMAGE's CH-byte grid, not the block's: every cache line of the
	 * image is then fetched by exactly one load (block-relative chunks straddle two lines
	 * each and fetched 1.5x the image).  sk = where the block starts inside its first
	 * chunk; staged position = payload position + sk. */
	const uint32_t sk = (do_parse || do_sum) ? (uint32_t)(b.src_off & (PS_CH - 1u)) : 0u;
	const uint32_t stage_len = (do_parse || do_sum) ? b.src_len + sk : 0u;	/* staged bytes, from the chunk grid */

	/* ---- loader set-up: which block and piece this lane fetches in instruction i ---- */
	ps_loader L;
	const uint32_t piece = lane % PS_PIECES;
#pragma unroll
	for (uint32_t k = 0; k < PS_PIECES; k++) {
		const int from = (int)(k * PS_BPI + lane / PS_PIECES);
		const uint64_t a_off = b.src_off - sk, a_room = room64 + sk;	/* both from the grid point before the block */
		const uint32_t off_lo = __shfl((uint32_t)a_off, from), off_hi = __shfl((uint32_t)(a_off >> 32), from);
		const uint32_t rm_lo = __shfl((uint32_t)a_room, from), rm_hi = __shfl((uint32_t)(a_room >> 32), from);
		L.ptr[k] = ((uint64_t)off_hi << 32) | off_lo;
		L.room[k] = ((uint64_t)rm_hi << 32) | rm_lo;
		L.need[k] = __shfl(stage_len, from);
	}
	uint32_t max_len = stage_len;
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) {
		const uint32_t o = __shfl_xor(max_len, m);
		max_len = o > max_len ? o : max_len;
	}
	const uint32_t nrounds = (uint32_t)(((uint64_t)max_len + PS_CH - 1) / PS_CH);

	/* store chunk c from registers into its ring slot */
#define PS_STORE_CHUNK(c_)                                                                        \
	do {                                                                                      \
		_Pragma("unroll")                                                                 \
		for (uint32_t k_ = 0; k_ < PS_PIECES; k_++) {                                     \
			const uint32_t slot_ = k_ * PS_BPI + lane / PS_PIECES;                    \
			const uint32_t j_ = (((c_) * (PS_CH / 4u)) + 4u * piece) & (PS_RINGW - 1u); \
			R[j_ + 0][slot_] = fl[k_].x;                                              \
			if (j_ == 0) {                                                            \
				R[PS_RINGW + 0][slot_] = fl[k_].x;                                \
				R[PS_RINGW + 1][slot_] = fl[k_].y;                                \
				R[PS_RINGW + 2][slot_] = fl[k_].z;                                \
				R[PS_RINGW + 3][slot_] = fl[k_].w;                                \
			}                                                                         \
			R[j_ + 1][slot_] = fl[k_].y;                                              \
			R[j_ + 2][slot_] = fl[k_].z;                                              \
			R[j_ + 3][slot_] = fl[k_].w;                                              \
		}                                                                                 \
	} while (0)

	uint4 fl[PS_PIECES];
	if (nrounds) {
		ps_load_chunk(src, L, 0, piece, fl);
		PS_STORE_CHUNK(0u);
		ps_load_chunk(src, L, 1, piece, fl);
		PS_STORE_CHUNK(1u);
		ps_load_chunk(src, L, 2, piece, fl);
	}

	/* ---- per-lane parse state ---- */
	const uint8_t *gs = src + b.src_off;
	const int iend = (int)b.src_len;
	const int oend = (int)b.dst_cap;
	const int glimit = room64 > 0x7fffffffull ? 0x7fffffff : (int)room64;	/* payload offsets >= glimit are outside the image */
	const int dict = (b.flags & LA_LZ4B_DEPENDENT) ? 65536 : 0;	/* lz4.c:563-584: any offset reaches the zero-filled prefix */
	const bool eligible = EMIT && have && la_lz4_fast_eligible(b);
	const bool emit = eligible && do_parse && table_off[i + 1] <= table_cap;
	uint64_t *tab = emit ? (uint64_t *)(table + table_off[i]) : nullptr;	/* slot is 64-byte aligned */
	int ip = 0, op = 0;
	uint32_t nseq = 0;
	bool ok = iend > 0;
	uint32_t done = (!do_parse || !ok) ? 1u : 0u;
	/* checksum state */
	uint32_t v1 = XXH_P1 + XXH_P2, v2 = XXH_P2, v3 = 0, v4 = 0u - XXH_P1;
	uint32_t hp = 0;	/* payload offset of the next stripe */
	bool hfin = false;

#define EMIT_SEQ(lit_src_, lit_len_, dst_, off_)                                                   \
	do {                                                                                       \
		stage[nseq & (PS_STAGE - 1u)][threadIdx.x] = (uint64_t)(uint16_t)(lit_src_) |      \
		    ((uint64_t)(uint16_t)(lit_len_) << 16) | ((uint64_t)(uint16_t)(dst_) << 32) |  \
		    ((uint64_t)(uint16_t)(off_) << 48);                                            \
	} while (0)
	uint32_t nfl = 0;	/* entries already written to the table (multiple of 8) */
	/* write every complete group of eight staged entries */
	auto flush_groups = [&]() {
		while (__ballot(emit && nseq - nfl >= 8u) != 0) {
			if (emit && nseq - nfl >= 8u) {
				const uint32_t g = nfl & (PS_STAGE - 1u);	/* 0 or 8 */
				uint4 *o4 = (uint4 *)(tab + nfl);
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const uint64_t a_ = stage[g + 2 * j][threadIdx.x], b_ = stage[g + 2 * j + 1][threadIdx.x];
					o4[j] = make_uint4((uint32_t)a_, (uint32_t)(a_ >> 32), (uint32_t)b_, (uint32_t)(b_ >> 32));
				}
				nfl += 8u;
			}
		}
	};

	uint32_t *const Rw = &R[0][0];	/* row stride 64 dwords; row PS_RINGW repeats row 0 */
	for (uint32_t r = 0; r < nrounds; r++) {
		const uint32_t base = r * PS_CH;
		const uint32_t rend = base + PS_CH;
		const uint32_t se32 = base + 2u * PS_CH;	/* ring holds [base, se32); no wrap: payloads stay below 2^31 */
		if (EMIT)
			flush_groups();

		/* one payload byte at offset p >= base: ring, or (length-extension runs and
		 * literal runs longer than the ring) the image itself */
		auto get_byte = [&](int p) -> uint32_t {
			const uint32_t sp = (uint32_t)p + sk;
			if (sp < se32)
				return (R[(sp >> 2) & (PS_RINGW - 1u)][lane] >> (8 * (sp & 3u))) & 0xffu;
			return p < glimit ? (uint32_t)gs[p] : 0u;
		};

		/* One sequence of this lane by the book: every rule, every rare case (runs of
		 * length-extension bytes, an offset beyond the staged bytes, the final
		 * literal-only sequence, failures).  The trip loop below only comes here for
		 * lanes its branch-free fast path has turned away. */
		auto careful_step = [&]() {
			const uint32_t token = get_byte(ip);
			int length = (int)(token >> 4);
			int p = ip + 1;
			bool fail = false;
			if (length == 15) {
				fail = p >= iend - 15;
				if (!fail) {
					uint32_t x;
					do {
						x = get_byte(p++);
						length += (int)x;
						/* a run of 0xff extension bytes cannot push the 32-bit sum around (a legacy
						 * block holds 8 MiB of them): longer than the output means failure below */
						if (p >= iend - 15 || length > oend)
							break;
					} while (x == 255);
				}
			}
			if (!fail && (op + length > oend - LZ4_MFLIMIT || p + length > iend - (2 + 1 + LZ4_LASTLIT))) {
				/* final, literal-only sequence: must consume the payload exactly */
				if (p + length != iend || op + length > oend)
					ok = false;
				else if (emit && length > 0) {
					EMIT_SEQ(p, length, op, 0);
					nseq++;
				}
				op += length;
				done = 1;
				return;
			}
			if (!fail) {
				const int lit_src = p, lit_len = length, lit_dst = op;
				p += length;
				const int opl = op + length;
				const int offset = (int)(get_byte(p) | (get_byte(p + 1) << 8));
				p += 2;
				length = (int)(token & 15);
				if (length == 15) {
					uint32_t x;
					do {
						x = get_byte(p++);
						length += (int)x;
						if (p >= iend - LZ4_LASTLIT + 1 || length > oend) { fail = true; break; }
					} while (x == 255);
				}
				length += 4;
				fail = fail || offset == 0 || offset > opl + dict || opl + length > oend - LZ4_LASTLIT;
				if (!fail) {
					if (emit)
						EMIT_SEQ(lit_src, lit_len, lit_dst, offset);
					nseq++;
					op = opl + length;
					ip = p;
				}
			}
			if (fail) {
				ok = false;
				done = 1;
			}
		};

		/* ---- parse: a lane takes one whole sequence per trip while its token is staged
		 * (it may run up to a chunk ahead); the round lasts until no lane is left inside
		 * chunk r.  This kernel is bound by instruction issue, so the trip is branch-free
		 * straight-line code on integer registers for the common sequence (no run of
		 * length-extension bytes beyond the first, offset staged, not the last of the
		 * block): two LDS round trips of two dwords each, every rule evaluated as a
		 * predicate, state committed with selects.  Lanes the predicates turn away take
		 * the careful step instead. ---- */
		for (;;) {
			const bool live = done == 0;
			if (__ballot(live && (uint32_t)ip + sk < rend) == 0)
				break;
			const bool full = EMIT && emit && nseq - nfl >= PS_STAGE;
			const uint32_t uip = (uint32_t)ip, uop = (uint32_t)op, sip = uip + sk;	/* sip: staged position of the token */
			const bool act = live && sip + 8u <= se32 && !full;
			const uint32_t row = (((sip >> 2) & (PS_RINGW - 1u)) << 6) | lane;
			const uint32_t w0 = __builtin_amdgcn_alignbit(Rw[row + 64], Rw[row], sip << 3);
			const uint32_t nib_l = (w0 >> 4) & 15u, nib_m = w0 & 15u, x1 = (w0 >> 8) & 0xffu;
			const bool ext1 = nib_l == 15u;
			const uint32_t ll = nib_l + (ext1 ? x1 : 0u);
			const uint32_t p = uip + (ext1 ? 2u : 1u);
			const bool s1 = ext1 && (x1 == 255u || (int)(uip + 1u) >= iend - 15);
			const uint32_t opl = uop + ll, q = p + ll;
			const bool lastseq = (int)opl > oend - LZ4_MFLIMIT || (int)q > iend - (2 + 1 + LZ4_LASTLIT);
			const uint32_t sq = q + sk;
			const uint32_t row2 = (((sq >> 2) & (PS_RINGW - 1u)) << 6) | lane;
			const uint32_t o3 = __builtin_amdgcn_alignbit(Rw[row2 + 64], Rw[row2], sq << 3);
			const uint32_t off = o3 & 0xffffu, x2 = (o3 >> 16) & 0xffu;
			const bool ext2 = nib_m == 15u;
			const uint32_t ml = nib_m + 4u + (ext2 ? x2 : 0u);
			const uint32_t p2 = q + (ext2 ? 3u : 2u);
			const bool s2 = ext2 && (x2 == 255u || (int)p2 >= iend - LZ4_LASTLIT + 1);
			const bool bad = off == 0u || (int)off > (int)opl + dict || (int)(opl + ml) > oend - LZ4_LASTLIT;
			/* offset not staged yet: a lane that runs ahead of the round simply waits for
			 * the next chunk; one still inside chunk r has a literal run longer than the
			 * ring and lets the careful step fetch the three bytes from the image */
			const bool far = sq + 4u > se32;
			const bool hold = far && sip >= rend;
			const bool slow = act && !hold && (s1 || lastseq || far || s2 || bad);
			const bool commit = act && !hold && !slow;
			if (EMIT && commit && emit)
				stage[nseq & (PS_STAGE - 1u)][threadIdx.x] =
				    (uint64_t)(p | (ll << 16)) | ((uint64_t)(uop | (o3 << 16)) << 32);
			nseq += commit ? 1u : 0u;
			op = commit ? (int)(opl + ml) : op;
			ip = commit ? (int)p2 : ip;
			if (__ballot(slow || (live && full)) != 0) {
				if (EMIT && __ballot(live && full) != 0)
					flush_groups();
				if (slow)
					careful_step();
			}
		}

		/* ---- block checksum: the XXH32 stripes that START in chunk r (a stripe is 16 payload
		 * bytes; on the image's chunk grid it may reach into chunk r+1, which is staged) ---- */
		if (SUMS && do_sum && !hfin) {
#pragma unroll
			for (uint32_t sidx = 0; sidx < PS_PIECES; sidx++) {
				const uint32_t sp = hp + sk;
				if (hp + 16u <= b.src_len && sp < rend) {
					const uint32_t rowh = (((sp >> 2) & (PS_RINGW - 1u)) << 6) | lane;
					const uint32_t d0 = Rw[rowh], d1 = Rw[rowh + 64], d2 = Rw[rowh + 128], d3 = Rw[rowh + 192], d4 = Rw[rowh + 256];
					const uint32_t sh = sp << 3;
					v1 = xxh_round(v1, __builtin_amdgcn_alignbit(d1, d0, sh));
					v2 = xxh_round(v2, __builtin_amdgcn_alignbit(d2, d1, sh));
					v3 = xxh_round(v3, __builtin_amdgcn_alignbit(d3, d2, sh));
					v4 = xxh_round(v4, __builtin_amdgcn_alignbit(d4, d3, sh));
					hp += 16u;
				}
			}
			if (hp + 16u > b.src_len && b.src_len + sk <= se32) {
				/* no whole stripe left and the rest is staged: merge, tail, avalanche
				 * (xxhash.c:268-291) */
				uint32_t h = b.src_len >= 16 ? rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18) : XXH_P5;
				h += b.src_len;
				uint32_t p = hp;
				for (; p + 4 <= b.src_len; p += 4) {
					const uint32_t sp = p + sk, rowh = (((sp >> 2) & (PS_RINGW - 1u)) << 6) | lane;
					h = rotl32(h + __builtin_amdgcn_alignbit(Rw[rowh + 64], Rw[rowh], sp << 3) * XXH_P3, 17) * XXH_P4;
				}
				for (; p < b.src_len; p++) {
					const uint32_t sp = p + sk;
					const uint32_t by = (R[(sp >> 2) & (PS_RINGW - 1u)][lane] >> (8 * (sp & 3u))) & 0xffu;
					h = rotl32(h + by * XXH_P5, 11) * XXH_P1;
				}
				h = xxh_avalanche(h);
				if (h != b.block_sum)
					sum_status[i] = LA_ST_LZ4_BAD_BLOCK_SUM;	/* outranks a decode failure: checked first, lz4.c:517 vs :594 */
				hfin = true;
			}
		}

		/* ---- chunk r is spent: chunk r+2 takes its slot, chunk r+3 goes in flight ---- */
		PS_STORE_CHUNK(r + 2u);
		ps_load_chunk(src, L, r + 3u, piece, fl);
	}
	if (SUMS && do_sum && !hfin && b.src_len == 0) {	/* (an empty payload on a grid point stages nothing) */
		if (xxh_avalanche(XXH_P5) != b.block_sum)
			sum_status[i] = LA_ST_LZ4_BAD_BLOCK_SUM;
	}
	if (!have)
		return;
	if (pre_failed) {	/* block already failed before this launch */
		out_len[i] = 0;
		nseq_out[i] = 0;
		return;
	}
	if (b.flags & LA_LZ4B_STORED) {
		out_len[i] = b.src_len;
		nseq_out[i] = 0;
		return;
	}
	if (done == 0)
		ok = false;	/* cannot happen: every chain ends inside its payload */
	if (EMIT)
		flush_groups();
	if (emit && ok)
		for (uint32_t j = nfl; j < nseq; j++)	/* the last, incomplete group of eight */
			tab[j] = stage[j & (PS_STAGE - 1u)][threadIdx.x];
#undef EMIT_SEQ
#undef PS_STORE_CHUNK
	out_len[i] = ok ? (uint32_t)op : 0u;
	/* an eligible block without a table slot must go to the general kernel */
	nseq_out[i] = ok ? ((eligible && !emit) ? 0xFFFFFFFFu : nseq) : 0u;
	if (!ok)
		status[i] = LA_ST_LZ4_DECODE;
}

void la_launch_lz4_parse_staged(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes,
    const la_lz4_block *d_blocks, uint32_t n, uint32_t *d_out_len, uint32_t *d_nseq,
    uint32_t *d_status, uint32_t *d_sum_status, la_lz4_seq *d_table, const uint64_t *d_table_off,
    uint64_t table_cap)
{
	if (n == 0) return;
	const dim3 grid((n + PS_THREADS - 1) / PS_THREADS), wg(PS_THREADS);
	if (d_table && d_sum_status)
		hipLaunchKernelGGL((lz4_parse_staged_kernel<true, true>), grid, wg, 0, s, d_src, src_bytes, d_blocks, n,
		    d_out_len, d_nseq, d_status, d_sum_status, d_table, d_table_off, table_cap);
	else if (d_table)
		hipLaunchKernelGGL((lz4_parse_staged_kernel<true, false>), grid, wg, 0, s, d_src, src_bytes, d_blocks, n,
		    d_out_len, d_nseq, d_status, d_sum_status, d_table, d_table_off, table_cap);
	else if (d_sum_status)
		hipLaunchKernelGGL((lz4_parse_staged_kernel<false, true>), grid, wg, 0, s, d_src, src_bytes, d_blocks, n,
		    d_out_len, d_nseq, d_status, d_sum_status, d_table, d_table_off, table_cap);
	else
		hipLaunchKernelGGL((lz4_parse_staged_kernel<false, false>), grid, wg, 0, s, d_src, src_bytes, d_blocks, n,
		    d_out_len, d_nseq, d_status, d_sum_status, d_table, d_table_off, table_cap);
}
