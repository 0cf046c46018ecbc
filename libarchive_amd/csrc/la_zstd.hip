/*
 * la_zstd.hip -- Zstandard frame decoder for gfx950, first cut: the data plane of the zstd read filter
 * (SURVEY section 8 f3).  Replaces, for a batch of whole frames per call, what the reference's filter gets from
 * libzstd's ZSTD_decompressStream (libarchive/archive_read_support_filter_zstd.c:171-260): frame header, raw / RLE /
 * compressed blocks (Huffman literals in 1 or 4 streams with direct or FSE-coded weights, FSE sequences with
 * predefined / RLE / described / repeated tables, repeat offsets), and the XXH64 content checksum (RFC 8878).
 *
 * Parallelism is ACROSS frames: a frame is one serial chain (every block may reach back into the previous ones,
 * entropy tables and repeat offsets carry over).  That is the shape of pzstd output and of seekable / chunked .zst
 * files; a one-frame .zst runs on one wave and is this design's worst case, like a one-member .gz.
 *
 *   zstd_frames_wave_kernel (default)  one WAVE per frame.  All 64 lanes run the same decoder on the same frame
 *       (uniform control flow: every lane computes the same header, table and sequence values, so nothing is
 *       broadcast and nothing diverges); the entropy tables live in LDS (10 KiB per wave), the bit streams are read
 *       through a 64-bit register window, and the byte moving is split over the lanes: raw / RLE blocks, literal
 *       runs and matches (an overlapping match reads position k mod offset of the bytes in front of it, so all
 *       its bytes go out at once), the four Huffman streams on four lanes, XXH64's four accumulators on four lanes.
 *       Sequences go 64 at a time: the wave decodes them (uniform, lane j keeps sequence j), then every lane copies its
 *       own literal run, one s_waitcnt vmcnt(0), and the matches follow in ballot rounds (a source that reaches into
 *       the group waits for the lanes whose sequences it touches, found with two lower bounds over the lanes' end
 *       positions; the scheme of la_lz4_wide.hip's in-order path).
 *   zstd_frames_kernel (LA_ZSTD_OPT_LANE_KERNEL)  the first form, one LANE per frame with its tables in an HBM
 *       workspace slot: same results, kept as a cross-check.
 * Measured (tools/measure_zstd.py, profiles/r02_zstd.txt): 16 384 frames of 64 KiB at level 3 decode at 20 GiB/s
 * resident in HBM (lane form: 6.1) against 2.9 GiB/s for libzstd on one host core.  87 % of a frame's cycles go to the
 * uniform 64-sequence decode loop: about 350 instructions per sequence, but ONE dependency chain per wave (state -> LDS
 * table word -> length table -> bit read -> next state); fewer branches changed nothing and moving the chain to the scalar
 * unit (v_readfirstlane, ZSTD_SCALAR) made it slower, and the 10.8 KiB of LDS tables hold a CU to 14 waves.  Next: two or
 * more frames interleaved per wave (independent chains), smaller tables for occupancy.
 */
#include "la_dev.h"

#define P64_1 11400714785074694791ULL
#define P64_2 14029467366897019727ULL
#define P64_3 1609587929392839161ULL
#define P64_4 9650029242287828579ULL
#define P64_5 2870177450012600261ULL
__device__ static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ static uint64_t rd64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ static uint32_t rd32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ static uint64_t xxh64_round(uint64_t acc, uint64_t in) { acc += in * P64_2; acc = rotl64(acc, 31); return acc * P64_1; }
__device__ static uint64_t xxh64_merge(uint64_t h, uint64_t v) { v = xxh64_round(0, v); h ^= v; return h * P64_1 + P64_4; }

/* XXH64 (the frame's content checksum is its low 32 bits, RFC 8878 3.1.1) */
__device__ static uint64_t dev_xxh64(const uint8_t *p, size_t len, uint64_t seed)
{
	const uint8_t *end = p + len;
	uint64_t h;
	if (len >= 32) {
		uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
		do {
			v1 = xxh64_round(v1, rd64(p)); v2 = xxh64_round(v2, rd64(p + 8));
			v3 = xxh64_round(v3, rd64(p + 16)); v4 = xxh64_round(v4, rd64(p + 24));
			p += 32;
		} while (p + 32 <= end);
		h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
		h = xxh64_merge(h, v1); h = xxh64_merge(h, v2); h = xxh64_merge(h, v3); h = xxh64_merge(h, v4);
	} else {
		h = seed + P64_5;
	}
	h += (uint64_t)len;
	while (p + 8 <= end) { h ^= xxh64_round(0, rd64(p)); h = rotl64(h, 27) * P64_1 + P64_4; p += 8; }
	if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P64_1; h = rotl64(h, 23) * P64_2 + P64_3; p += 4; }
	while (p < end) { h ^= (uint64_t)(*p++) * P64_5; h = rotl64(h, 11) * P64_1; }
	h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
	return h;
}

__device__ static void dev_copy(uint8_t *d, const uint8_t *s, size_t n)
{
	size_t i = 0;
	for (; i + 8 <= n; i += 8) { uint64_t v; __builtin_memcpy(&v, s + i, 8); __builtin_memcpy(d + i, &v, 8); }
	for (; i < n; i++) d[i] = s[i];
}
__device__ static void dev_fill(uint8_t *d, uint8_t v, size_t n)
{
	const uint64_t w = 0x0101010101010101ull * v;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) __builtin_memcpy(d + i, &w, 8);
	for (; i < n; i++) d[i] = v;
}

/* ---- wave form: the same decoder run by all 64 lanes of a wave on ONE frame (uniform control flow, every lane
 * computes the same header / table / sequence values), with the byte moving split over the lanes ---- */
__device__ static void wave_fence() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <bool W> __device__ __forceinline__ static void t_copy(uint8_t *d, const uint8_t *s, size_t n)
{
	if (!W) { dev_copy(d, s, n); return; }
	const size_t lane = __lane_id(), body = n & ~(size_t)7;
	for (size_t i = lane * 8; i < body; i += 512) { uint64_t v; __builtin_memcpy(&v, s + i, 8); __builtin_memcpy(d + i, &v, 8); }
	for (size_t i = body + lane; i < n; i += 64) d[i] = s[i];
}
template <bool W> __device__ __forceinline__ static void t_fill(uint8_t *d, uint8_t v, size_t n)
{
	if (!W) { dev_fill(d, v, n); return; }
	const uint64_t w = 0x0101010101010101ull * v;
	const size_t lane = __lane_id(), body = n & ~(size_t)7;
	for (size_t i = lane * 8; i < body; i += 512) __builtin_memcpy(d + i, &w, 8);
	for (size_t i = body + lane; i < n; i += 64) d[i] = v;
}
/* match of ml bytes at dst[out..] from offset bytes back; every source byte of an overlapping match (offset < ml)
 * is one of the `offset` bytes in front of it, so the wave form copies all positions at once */
template <bool W> __device__ __forceinline__ static void t_match(uint8_t *dst, size_t out, uint32_t offset, uint32_t ml)
{
	if (!W) {
		for (uint32_t k = 0; k < ml; k++) dst[out + k] = dst[out + k - offset];
		return;
	}
	const uint32_t lane = __lane_id();
	const uint8_t *s = dst + out - offset;
	if (offset >= ml) {
		for (uint32_t k = lane; k < ml; k += 64) dst[out + k] = s[k];
	} else {
		for (uint32_t k = lane; k < ml; k += 64) dst[out + k] = s[k % offset];
	}
}
/* XXH64, wave form: lane j & 3 runs accumulator j over the 32-byte stripes, the rest is uniform */
__device__ __forceinline__ static uint64_t wave_xxh64(const uint8_t *p, size_t len, uint64_t seed)
{
	const uint8_t *end = p + len;
	uint64_t h;
	if (len >= 32) {
		const uint32_t j = __lane_id() & 3u;
		uint64_t v = j == 0 ? seed + P64_1 + P64_2 : (j == 1 ? seed + P64_2 : (j == 2 ? seed : seed - P64_1));
		const size_t stripes = len / 32;
		const uint8_t *q = p + 8 * j;
#pragma unroll 8
		for (size_t s = 0; s < stripes; s++)
			v = xxh64_round(v, rd64(q + 32 * s));
		uint64_t a[4];
		for (int k = 0; k < 4; k++) {
			const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, k, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), k, 64);
			a[k] = (uint64_t)lo | ((uint64_t)hi << 32);
		}
		h = rotl64(a[0], 1) + rotl64(a[1], 7) + rotl64(a[2], 12) + rotl64(a[3], 18);
		h = xxh64_merge(h, a[0]); h = xxh64_merge(h, a[1]); h = xxh64_merge(h, a[2]); h = xxh64_merge(h, a[3]);
		p += stripes * 32;
	} else {
		h = seed + P64_5;
	}
	h += (uint64_t)len;
	while (p + 8 <= end) { h ^= xxh64_round(0, rd64(p)); h = rotl64(h, 27) * P64_1 + P64_4; p += 8; }
	if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P64_1; h = rotl64(h, 23) * P64_2 + P64_3; p += 4; }
	while (p < end) { h ^= (uint64_t)(*p++) * P64_5; h = rotl64(h, 11) * P64_1; }
	h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
	return h;
}

/* ---- bit readers ---- */
/* n (<= 32) bits at bit position pos of the little-endian bit array src[0..len); positions outside read as zero */
__device__ static uint32_t bits_at(const uint8_t *src, size_t len, int64_t pos, unsigned n)
{
	uint64_t v = 0;
	if (n == 0) return 0;
	for (int i = 0; i < 6; i++) {	/* up to 6 bytes cover 32 bits at any bit phase */
		int64_t byte = (pos >> 3) + i;	/* arithmetic shift: floor for negative positions */
		uint64_t b = (byte >= 0 && (uint64_t)byte < len) ? src[byte] : 0;
		v |= b << (8 * i);
	}
	v >>= (unsigned)(pos & 7);
	return (uint32_t)(v & ((n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1)));
}
__device__ static int highbit(uint32_t v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

/* backward stream: returns the bit position just below the end marker, or -1 when the last byte is zero */
__device__ static int64_t rev_init(const uint8_t *src, size_t len)
{
	if (len == 0 || src[len - 1] == 0) return -1;
	return (int64_t)(len - 1) * 8 + highbit(src[len - 1]);
}
__device__ static uint32_t rev_read(const uint8_t *src, size_t len, int64_t *pos, unsigned n)
{
	*pos -= n;
	return bits_at(src, len, *pos, n);
}

/* ---- FSE ---- */
typedef struct { uint8_t sym, nbits; uint16_t base; } fse_ent;
typedef struct { fse_ent e[512]; int al; } fse_tab;

/* normalized counts (RFC 8878 4.1.1); returns bytes consumed or -1 */
__device__ static int fse_read_ncount(const uint8_t *src, size_t len, int max_al, int max_sym, int16_t *norm, int *n_sym, int *al_out)
{
	int64_t bp = 0;
	if (len == 0) return -1;
	const int al = (int)bits_at(src, len, bp, 4) + 5; bp += 4;
	if (al > max_al) return -1;
	int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0;
	while (remaining > 1 && sym <= max_sym) {
		if ((size_t)((bp + 7) >> 3) > len + 4) return -1;
		const int max = (2 * threshold - 1) - remaining;
		int count;
		const uint32_t v = bits_at(src, len, bp, (unsigned)nbits);
		if ((int)(v & (uint32_t)(threshold - 1)) < max) {
			count = (int)(v & (uint32_t)(threshold - 1));
			bp += nbits - 1;
		} else {
			count = (int)(v & (uint32_t)(2 * threshold - 1));
			if (count >= threshold) count -= max;
			bp += nbits;
		}
		count--;	/* -1 = "less than one" */
		remaining -= count < 0 ? -count : count;
		norm[sym++] = (int16_t)count;
		if (count == 0) {	/* repeat flags: runs of zero probabilities */
			for (;;) {
				const uint32_t r = bits_at(src, len, bp, 2); bp += 2;
				for (uint32_t i = 0; i < r; i++) { if (sym > max_sym) return -1; norm[sym++] = 0; }
				if (r != 3) break;
			}
		}
		if (remaining < 1) return -1;
		while (remaining < threshold) { nbits--; threshold >>= 1; }
	}
	if (remaining != 1 || sym > max_sym + 1) return -1;
	const size_t used = (size_t)((bp + 7) >> 3);
	if (used > len) return -1;
	*n_sym = sym; *al_out = al;
	return (int)used;
}

__device__ static int fse_build(fse_tab *t, const int16_t *norm, int n_sym, int al)
{
	const int size = 1 << al;
	uint16_t next[256];
	int high = size - 1;
	t->al = al;
	for (int s = 0; s < n_sym; s++) {
		if (norm[s] == -1) { t->e[high--].sym = (uint8_t)s; next[s] = 1; }
		else next[s] = (uint16_t)norm[s];
	}
	const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
	int pos = 0;
	for (int s = 0; s < n_sym; s++)
		for (int i = 0; i < norm[s]; i++) {
			t->e[pos].sym = (uint8_t)s;
			do { pos = (pos + step) & mask; } while (pos > high);
		}
	if (pos != 0) return -1;
	for (int u = 0; u < size; u++) {
		const int s = t->e[u].sym;
		const int nx = next[s]++;
		const int nb = al - highbit((uint32_t)nx);
		t->e[u].nbits = (uint8_t)nb;
		t->e[u].base = (uint16_t)((nx << nb) - size);
	}
	return 0;
}
__device__ static void fse_rle(fse_tab *t, int sym) { t->al = 0; t->e[0].sym = (uint8_t)sym; t->e[0].nbits = 0; t->e[0].base = 0; }

/* ---- Huffman (RFC 8878 4.2) ---- */
typedef struct { uint8_t sym[2048], nbits[2048]; int maxbits; } huf_tab;

__device__ static int huf_read(huf_tab *h, const uint8_t *src, size_t len)	/* returns bytes consumed or -1 */
{
	uint8_t w[256];
	int n = 0;
	size_t used;
	if (len < 1) return -1;
	const int hb = src[0];
	if (hb >= 128) {	/* direct: 4-bit weights */
		n = hb - 127;
		used = 1 + (size_t)(n + 1) / 2;
		if (used > len) return -1;
		for (int i = 0; i < n; i++)
			w[i] = (i & 1) ? (src[1 + i / 2] & 15) : (src[1 + i / 2] >> 4);
	} else {		/* FSE-compressed weights, two interleaved states */
		used = 1 + (size_t)hb;
		if (hb == 0 || used > len) return -1;
		int16_t norm[16]; int ns, al;
		fse_tab t;
		const int c = fse_read_ncount(src + 1, (size_t)hb, 6, 11, norm, &ns, &al);
		if (c < 0 || fse_build(&t, norm, ns, al) < 0) return -1;
		const uint8_t *bs = src + 1 + c; const size_t bl = (size_t)hb - (size_t)c;
		int64_t pos = rev_init(bs, bl);
		if (pos < 0) return -1;
		uint32_t s1 = rev_read(bs, bl, &pos, (unsigned)al), s2 = rev_read(bs, bl, &pos, (unsigned)al);
		if (pos < 0) return -1;
		for (;;) {
			if (n > 253) return -1;
			w[n++] = t.e[s1].sym;
			s1 = t.e[s1].base + rev_read(bs, bl, &pos, t.e[s1].nbits);
			if (pos < 0) { w[n++] = t.e[s2].sym; break; }
			if (n > 253) return -1;
			w[n++] = t.e[s2].sym;
			s2 = t.e[s2].base + rev_read(bs, bl, &pos, t.e[s2].nbits);
			if (pos < 0) { w[n++] = t.e[s1].sym; break; }
		}
	}
	/* the last weight completes a power of two */
	uint32_t sum = 0;
	for (int i = 0; i < n; i++) { if (w[i] > 11) return -1; if (w[i]) sum += 1u << (w[i] - 1); }
	if (sum == 0) return -1;
	const int maxbits = highbit(sum) + 1;
	if (maxbits > 11) return -1;
	const uint32_t left = (1u << maxbits) - sum;
	if (left == 0 || (left & (left - 1))) return -1;
	w[n++] = (uint8_t)(highbit(left) + 1);
	h->maxbits = maxbits;
	uint32_t pos = 0;
	for (int wt = 1; wt <= maxbits; wt++)
		for (int s = 0; s < n; s++)
			if (w[s] == wt) {
				const uint32_t cnt = 1u << (wt - 1);
				for (uint32_t i = 0; i < cnt; i++) { h->sym[pos + i] = (uint8_t)s; h->nbits[pos + i] = (uint8_t)(maxbits + 1 - wt); }
				pos += cnt;
			}
	if (pos != (1u << maxbits)) return -1;
	return (int)used;
}

/* The wave kernel runs the decoder uniformly on all lanes, but the compiler cannot know that values loaded from
 * memory are the same in every lane and would keep them in vector registers (every `if` an exec-mask dance, all
 * arithmetic on the vector unit).  uni<true>() moves such a value to a scalar register (v_readfirstlane): the bit
 * reader, the FSE states and the sequence values then live on the scalar unit with real branches.  uni<false>() is
 * the identity for the lane kernel and for the four per-lane Huffman streams. */
#ifndef ZSTD_SCALAR
#define ZSTD_SCALAR false	/* measured: 63.4 ms with the uniform values moved to scalar registers, 57 ms without */
#endif
template <bool U> __device__ __forceinline__ static uint32_t uni(uint32_t v) { return U ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v; }
template <bool U> __device__ __forceinline__ static int32_t unis(int32_t v) { return U ? __builtin_amdgcn_readfirstlane(v) : v; }
template <bool U> __device__ __forceinline__ static uint64_t uni64(uint64_t v)
{
	return U ? ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
	    ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32)) : v;
}

/* windowed reader of a backward stream: 64 bits of the stream in a register, one unaligned 8-byte load per refill
 * (the byte-wise bits_at above costs six dependent-latency loads per read) */
struct rbits { const uint8_t *src; uint32_t len; uint64_t win; int32_t lo; };
__device__ __forceinline__ static void rb_init(rbits &b, const uint8_t *src, size_t len) { b.src = src; b.len = (uint32_t)len; b.win = 0; b.lo = 0x40000000; }
/* n (<= 32) bits at position p (may be negative: zero bits), p + n <= 8 * len; a stream is at most one block (128 KiB),
 * so positions are 32-bit */
template <bool U> __device__ __forceinline__ static uint32_t rb_at(rbits &b, int32_t p, unsigned n)
{
	p = unis<U>(p); n = uni<U>(n);
	/* one (rarely taken) branch per read; n = 0 reads as 0 through the empty mask */
	if ((p < b.lo) | (p + (int32_t)n > b.lo + 64)) {
		const int32_t hi_byte = (p + (int32_t)n + 7) >> 3, lo_byte = hi_byte - 8;	/* the window ends just above the bits asked for */
		if (lo_byte >= 0 && (uint32_t)hi_byte <= b.len) {
			b.win = uni64<U>(rd64(b.src + lo_byte));
		} else {
			uint64_t v = 0;
			for (int i = 0; i < 8; i++) {
				const int32_t byte = lo_byte + i;
				if (byte >= 0 && (uint32_t)byte < b.len) v |= (uint64_t)b.src[byte] << (8 * i);
			}
			b.win = uni64<U>(v);
		}
		b.lo = lo_byte * 8;
	}
	return (uint32_t)(b.win >> ((unsigned)(p - b.lo) & 63u)) & (uint32_t)((1ull << n) - 1ull);
}

/* (a wave-wide form of this reader -- 512 bytes of the stream in one register pair per lane, reads through v_readlane --
 * was measured slower: 68.8 ms against 58.3 on 16 384 frames; the kernel is bound by instruction issue, not by the refills) */
__device__ __forceinline__ static void bits_init(rbits &b, const uint8_t *s, size_t l) { rb_init(b, s, l); }
template <bool U> __device__ __forceinline__ static uint32_t bits_read(rbits &b, int32_t *pos, unsigned n) { *pos -= (int32_t)n; return rb_at<U>(b, *pos, n); }

__device__ __forceinline__ static int huf_stream(const huf_tab *h, const uint8_t *src, size_t len, uint8_t *out, size_t n)
{
	int32_t pos = (int32_t)rev_init(src, len);
	if (pos < 0) return -1;
	rbits b;
	rb_init(b, src, len);
	const unsigned mb = (unsigned)h->maxbits;
	for (size_t i = 0; i < n; i++) {
		const uint32_t idx = rb_at<false>(b, pos - (int32_t)mb, mb);
		out[i] = h->sym[idx];
		pos -= h->nbits[idx];
		if (pos < 0) return -1;
	}
	return pos == 0 ? 0 : -1;
}

/* ---- sequences ---- */
/* literal-length and match-length codes: baselines and extra bits (RFC 8878 3.1.1.3.2.1.1) */
struct seq_tabs { uint32_t ll_base[36]; uint32_t ml_base[53]; uint8_t ll_bits[36]; uint8_t ml_bits[53]; };
__device__ static const seq_tabs SEQ_TABS = {
	{ 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536 },
	{ 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539 },
	{ 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 },
	{ 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 }
};
__device__ static const int16_t LL_DEF[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
__device__ static const int16_t ML_DEF[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
__device__ static const int16_t OF_DEF[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

typedef struct {
	huf_tab huf; int have_huf;
	fse_tab ll, of, ml; int have_ll, have_of, have_ml;
	uint32_t rep[3];
	uint8_t *lit;	/* 128 KiB + slack */
	seq_tabs tabs;	/* length code tables, a copy per frame state (LDS in the wave kernel: ds_read instead of flat loads through a pointer) */
} zframe;

/* one table of the sequences section; returns bytes consumed or -1 */
__device__ static int seq_table(fse_tab *t, int *have, int mode, const uint8_t *src, size_t len, int max_al, int max_sym,
    const int16_t *def, int def_n, int def_al)
{
	if (mode == 0) { if (fse_build(t, def, def_n, def_al) < 0) return -1; *have = 1; return 0; }
	if (mode == 1) { if (len < 1 || src[0] > max_sym) return -1; fse_rle(t, src[0]); *have = 1; return 1; }
	if (mode == 2) {
		int16_t norm[64]; int ns, al;
		const int c = fse_read_ncount(src, len, max_al, max_sym, norm, &ns, &al);
		if (c < 0 || fse_build(t, norm, ns, al) < 0) return -1;
		*have = 1;
		return c;
	}
	return *have ? 0 : -1;	/* repeat */
}

#define ZBLOCK_MAX (128u * 1024u)

/* one compressed block; returns bytes produced or -1 */
/* sequence decoder state: bit window, the three FSE states, the repeat offsets (registers, not the frame struct) */
template <class R> struct seqdec { R rb; int32_t pos; uint32_t sl, so, sm, r0, r1, r2; };
__device__ __forceinline__ static uint32_t fse_word(const fse_tab *t, uint32_t s) { uint32_t v; __builtin_memcpy(&v, &t->e[s], 4); return v; }	/* sym | nbits << 8 | base << 16 */

/* next sequence (RFC 8878 3.1.1.3.2.1.1): values, repeat-offset rule, state update unless it is the block's last */
template <bool U, class R> __device__ __forceinline__ static int seq_next(const zframe *f, seqdec<R> &d, bool last, uint32_t &ll, uint32_t &ml, uint32_t &offset)
{
	/* straight-line: the checks are collected in `bad` and looked at once (the loop is bound by instruction issue and
	 * every early return is a branch) */
	const uint32_t wo = uni<U>(fse_word(&f->of, d.so)), wm = uni<U>(fse_word(&f->ml, d.sm)), wl = uni<U>(fse_word(&f->ll, d.sl));
	uint32_t oc = wo & 0xFF, mc = wm & 0xFF, lc = wl & 0xFF;
	bool bad = (oc > 31) | (mc > 52) | (lc > 35);
	oc = oc > 31 ? 31 : oc; mc = mc > 52 ? 52 : mc; lc = lc > 35 ? 35 : lc;
	const uint32_t ov = (1u << oc) + bits_read<U>(d.rb, &d.pos, oc);	/* (code 0: value 1, no bits) */
	/* codes without extra bits (match lengths 3..34, literal lengths 0..15: the common case) need no table: the second
	 * LDS trip of the chain is skipped for them.  The extra bits of the match length and of the literal length come
	 * out of ONE read (at most 16 + 16 bits; the match length's were written last, so they are the high part). */
	uint32_t mb = mc + 3u, lb = lc, mbits = 0, lbits = 0;
	if (mc >= 32u) { mb = uni<U>(f->tabs.ml_base[mc]); mbits = uni<U>(f->tabs.ml_bits[mc]); }
	if (lc >= 16u) { lb = uni<U>(f->tabs.ll_base[lc]); lbits = uni<U>(f->tabs.ll_bits[lc]); }
	const uint32_t ev = bits_read<U>(d.rb, &d.pos, mbits + lbits);
	ml = mb + (ev >> lbits);
	ll = lb + (ev & ((1u << lbits) - 1u));
	bad |= d.pos < 0;
	/* repeat offsets (RFC 8878 3.1.1.5) with selects */
	const bool rep = ov <= 3;
	const uint32_t idx = ov - 1 + (ll == 0 ? 1u : 0u);	/* 0..3 when rep */
	const uint32_t cand = idx == 0 ? d.r0 : (idx == 1 ? d.r1 : (idx == 2 ? d.r2 : d.r0 - 1u));
	offset = rep ? cand : ov - 3;
	bad |= offset == 0;
	const bool shift = !rep | (idx != 0);	/* the history changes */
	const uint32_t n2 = (rep & (idx == 1)) ? d.r2 : d.r1;
	d.r2 = shift ? n2 : d.r2;
	d.r1 = shift ? d.r0 : d.r1;
	d.r0 = shift ? offset : d.r0;
	if (!last) {
		/* the three state updates out of ONE read (at most 9 + 9 + 8 bits; order in the stream: LL, ML, OF) */
		const uint32_t nl = (wl >> 8) & 0xFF, nm = (wm >> 8) & 0xFF, no = (wo >> 8) & 0xFF;
		const uint32_t sv = bits_read<U>(d.rb, &d.pos, nl + nm + no);
		d.sl = (wl >> 16) + (sv >> (nm + no));
		d.sm = (wm >> 16) + ((sv >> no) & ((1u << nm) - 1u));
		d.so = (wo >> 16) + (sv & ((1u << no) - 1u));
		bad |= d.pos < 0;
	}
	return bad ? -1 : 0;
}

template <bool W> __device__ __forceinline__ static int64_t zstd_block(zframe *f, const uint8_t *src, size_t len, uint8_t *dst, size_t dst_pos, size_t dst_cap)
{
	if (len < 1) return -1;	/* (libzstd: a compressed block needs at least a literals header) */
	/* ---- literals section ---- */
	const int ltype = src[0] & 3, sf = (src[0] >> 2) & 3;
	size_t hl, regen, comp = 0;
	int streams = 1;
	if (ltype < 2) {
		if (sf == 0 || sf == 2) { hl = 1; regen = src[0] >> 3; }
		else if (sf == 1) { if (len < 2) return -1; hl = 2; regen = (src[0] >> 4) | ((size_t)src[1] << 4); }
		else { if (len < 3) return -1; hl = 3; regen = (src[0] >> 4) | ((size_t)src[1] << 4) | ((size_t)src[2] << 12); }
	} else {
		if (sf < 2) {
			if (len < 3) return -1;
			hl = 3; streams = sf == 0 ? 1 : 4;
			const uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16);
			regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF;
		} else if (sf == 2) {
			if (len < 4) return -1;
			hl = 4; streams = 4;
			const uint32_t v = rd32(src);
			regen = (v >> 4) & 0x3FFF; comp = v >> 18;
		} else {
			if (len < 5) return -1;
			hl = 5; streams = 4;
			const uint64_t v = (uint64_t)rd32(src) | ((uint64_t)src[4] << 32);
			regen = (size_t)((v >> 4) & 0x3FFFF); comp = (size_t)(v >> 22);
		}
	}
	if (regen > ZBLOCK_MAX) return -1;
	const uint8_t *p = src + hl;
	size_t left = len - hl;
	if (ltype == 0) { if (regen > left) return -1; t_copy<W>(f->lit, p, regen); p += regen; left -= regen; }
	else if (ltype == 1) { if (left < 1) return -1; t_fill<W>(f->lit, p[0], regen); p += 1; left -= 1; }
	else {
		if (comp > left) return -1;
		const uint8_t *hp = p; size_t hleft = comp;
		if (ltype == 2) {
			const int c = huf_read(&f->huf, hp, hleft);
			if (c < 0) return -1;
			f->have_huf = 1; hp += c; hleft -= (size_t)c;
		} else if (!f->have_huf) return -1;
		if (streams == 1) {
			if (!W) {
				if (huf_stream(&f->huf, hp, hleft, f->lit, regen) < 0) return -1;
			} else {
				int bad = 0;
				if (__lane_id() == 0) bad = huf_stream(&f->huf, hp, hleft, f->lit, regen) < 0;
				if (__ballot(bad) != 0) return -1;
			}
		} else {
			if (hleft < 6) return -1;
			const size_t s1 = hp[0] | ((size_t)hp[1] << 8), s2 = hp[2] | ((size_t)hp[3] << 8), s3 = hp[4] | ((size_t)hp[5] << 8);
			if (6 + s1 + s2 + s3 > hleft) return -1;
			const size_t s4 = hleft - 6 - s1 - s2 - s3, q = (regen + 3) / 4;
			if (3 * q > regen) return -1;
			hp += 6;
			if (!W) {
				if (huf_stream(&f->huf, hp, s1, f->lit, q) < 0) return -1;
				if (huf_stream(&f->huf, hp + s1, s2, f->lit + q, q) < 0) return -1;
				if (huf_stream(&f->huf, hp + s1 + s2, s3, f->lit + 2 * q, q) < 0) return -1;
				if (huf_stream(&f->huf, hp + s1 + s2 + s3, s4, f->lit + 3 * q, regen - 3 * q) < 0) return -1;
			} else {	/* the four streams on four lanes */
				const uint32_t ln = __lane_id();
				int bad = 0;
				if (ln < 4) {
					const size_t so = ln == 0 ? 0 : (ln == 1 ? s1 : (ln == 2 ? s1 + s2 : s1 + s2 + s3));
					const size_t sl_ = ln == 0 ? s1 : (ln == 1 ? s2 : (ln == 2 ? s3 : s4));
					bad = huf_stream(&f->huf, hp + so, sl_, f->lit + ln * q, ln == 3 ? regen - 3 * q : q) < 0;
				}
				if (__ballot(bad) != 0) return -1;
			}
		}
		p += comp; left -= comp;
	}
	/* ---- sequences section ---- */
	if (W) wave_fence();	/* the literals are in the buffer */
	if (left < 1) return -1;
	size_t nseq = p[0];
	if (nseq == 0) { p += 1; left -= 1; }
	else if (nseq < 128) { p += 1; left -= 1; }
	else if (nseq < 255) { if (left < 2) return -1; nseq = ((nseq - 128) << 8) + p[1]; p += 2; left -= 2; }
	else { if (left < 3) return -1; nseq = p[1] + ((size_t)p[2] << 8) + 0x7F00; p += 3; left -= 3; }
	size_t out = dst_pos, lit_pos = 0;
	if (nseq) {
		if (left < 1) return -1;
		const int modes = p[0];
		/* (bits 0-1 are reserved; libzstd 1.4.8 ZSTD_decodeSeqHeaders does not look at them) */
		p += 1; left -= 1;
		int c;
		c = seq_table(&f->ll, &f->have_ll, modes >> 6, p, left, 9, 35, LL_DEF, 36, 6); if (c < 0) return -1; p += c; left -= (size_t)c;
		c = seq_table(&f->of, &f->have_of, (modes >> 4) & 3, p, left, 8, 31, OF_DEF, 29, 5); if (c < 0) return -1; p += c; left -= (size_t)c;
		c = seq_table(&f->ml, &f->have_ml, (modes >> 2) & 3, p, left, 9, 52, ML_DEF, 53, 6); if (c < 0) return -1; p += c; left -= (size_t)c;
		seqdec<rbits> sd;
		sd.pos = (int32_t)rev_init(p, left);
		if (sd.pos < 0) return -1;
		bits_init(sd.rb, p, left);
		sd.pos = unis<W && ZSTD_SCALAR>(sd.pos);
		sd.sl = bits_read<W && ZSTD_SCALAR>(sd.rb, &sd.pos, (unsigned)f->ll.al);
		sd.so = bits_read<W && ZSTD_SCALAR>(sd.rb, &sd.pos, (unsigned)f->of.al);
		sd.sm = bits_read<W && ZSTD_SCALAR>(sd.rb, &sd.pos, (unsigned)f->ml.al);
		if (sd.pos < 0) return -1;
		sd.r0 = uni<W && ZSTD_SCALAR>(f->rep[0]); sd.r1 = uni<W && ZSTD_SCALAR>(f->rep[1]); sd.r2 = uni<W && ZSTD_SCALAR>(f->rep[2]);
		if constexpr (!W) {
			for (size_t i = 0; i < nseq; i++) {
				uint32_t ll, ml, offset;
				if (seq_next<false>(f, sd, i + 1 == nseq, ll, ml, offset) < 0) return -1;
				if (ll > regen - lit_pos) return -1;
				if (out - dst_pos + ll + ml > ZBLOCK_MAX) return -1;
				if (out + ll + ml > dst_cap) return -2;
				dev_copy(dst + out, f->lit + lit_pos, ll); out += ll; lit_pos += ll;
				if (offset > out) return -1;
				t_match<false>(dst, out, offset, ml);
				out += ml;
			}
		} else {
			/* 64 sequences at a time: decoded by the whole wave (uniform), then executed one lane per sequence */
			const uint32_t lane = __lane_id();
			nseq = (size_t)uni64<ZSTD_SCALAR>(nseq); out = (size_t)uni64<ZSTD_SCALAR>(out); regen = (size_t)uni64<ZSTD_SCALAR>(regen);
			dst_cap = (size_t)uni64<ZSTD_SCALAR>(dst_cap); dst_pos = (size_t)uni64<ZSTD_SCALAR>(dst_pos);
			for (size_t base = 0; base < nseq; base += 64) {
				const uint32_t cnt = nseq - base < 64 ? (uint32_t)(nseq - base) : 64u;
				uint32_t my_ll = 0, my_ml = 0, my_off = 1;
				size_t my_out = out, my_lit = 0;
				for (uint32_t j = 0; j < cnt; j++) {
					uint32_t ll, ml, offset;
					if (ZSTD_SCALAR) {	/* loop-carried state back into scalar registers: the compiler cannot prove it uniform across the back edge */
						j = uni<ZSTD_SCALAR>(j);
						sd.pos = unis<ZSTD_SCALAR>(sd.pos); sd.rb.lo = unis<ZSTD_SCALAR>(sd.rb.lo); sd.rb.win = uni64<ZSTD_SCALAR>(sd.rb.win);
						sd.rb.len = uni<ZSTD_SCALAR>(sd.rb.len); sd.rb.src = (const uint8_t *)uni64<ZSTD_SCALAR>((uint64_t)sd.rb.src);
						sd.sl = uni<ZSTD_SCALAR>(sd.sl); sd.sm = uni<ZSTD_SCALAR>(sd.sm); sd.so = uni<ZSTD_SCALAR>(sd.so);
						sd.r0 = uni<ZSTD_SCALAR>(sd.r0); sd.r1 = uni<ZSTD_SCALAR>(sd.r1); sd.r2 = uni<ZSTD_SCALAR>(sd.r2);
						out = (size_t)uni64<ZSTD_SCALAR>(out); lit_pos = (size_t)uni64<ZSTD_SCALAR>(lit_pos);
					}
					if (seq_next<ZSTD_SCALAR>(f, sd, base + j + 1 == nseq, ll, ml, offset) < 0) return -1;
					if (ll > regen - lit_pos) return -1;
					if (out - dst_pos + ll + ml > ZBLOCK_MAX) return -1;
					if (out + ll + ml > dst_cap) return -2;
					if (offset > out + ll) return -1;
					if (lane == j) { my_ll = ll; my_ml = ml; my_off = offset; my_out = out; my_lit = lit_pos; }
					out += (size_t)ll + ml; lit_pos += ll;
				}
				const bool have = lane < cnt;
				/* literal runs: every lane its own */
				{
					const uint8_t *ls = f->lit + my_lit;
					uint8_t *ld = dst + my_out;
					uint32_t i = 0;
					for (; i + 8 <= my_ll; i += 8) { uint64_t v; __builtin_memcpy(&v, ls + i, 8); __builtin_memcpy(ld + i, &v, 8); }
					for (; i < my_ll; i++) ld[i] = ls[i];
				}
				wave_fence();	/* the literals of the group and everything in front of it are in place */
				/* matches: a source that reaches into the group waits for the lanes whose sequences it touches */
				const size_t g0 = (size_t)__shfl((int)(uint32_t)my_out, 0, 64) | ((size_t)__shfl((int)(uint32_t)((uint64_t)my_out >> 32), 0, 64) << 32);
				const size_t mdst = my_out + my_ll;
				const size_t s0 = mdst - my_off;
				const uint32_t span = my_ml < my_off ? my_ml : my_off;
				bool pendm = have && my_ml != 0;
				uint64_t depmask = 0;
				{
					/* ends of the group's sequences relative to its first byte (increasing over the lanes) */
					const uint32_t end = have ? (uint32_t)(mdst + my_ml - g0) : 0xFFFFFFFFu;
					const bool inside = pendm && s0 + span > g0;
					const uint32_t rlo = s0 > g0 ? (uint32_t)(s0 - g0) : 0u;
					const uint32_t rhi = inside ? (uint32_t)(s0 + span - 1 - g0) : 0u;
					uint32_t jlo = 0, jhi = 0;
#pragma unroll
					for (uint32_t bit = 32; bit; bit >>= 1) {
						const uint32_t e_lo = (uint32_t)__shfl((int)end, (int)(jlo + bit - 1u), 64);
						if (e_lo <= rlo) jlo += bit;
						const uint32_t e_hi = (uint32_t)__shfl((int)end, (int)(jhi + bit - 1u), 64);
						if (e_hi <= rhi) jhi += bit;
					}
					if (inside && lane != 0) {
						const uint32_t hi = jhi < lane ? jhi : lane - 1u;	/* own literals are in place */
						if (jlo <= hi) {
							const uint64_t upto = hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull);
							depmask = upto & ~((1ull << jlo) - 1ull);
						}
					}
				}
				for (;;) {
					const uint64_t pending = __ballot(pendm);
					if (pending == 0)
						break;
					if (pendm && (depmask & pending) == 0) {
						const uint8_t *ms = dst + s0;
						uint8_t *md = dst + mdst;
						if (my_off >= my_ml) {
							uint32_t i = 0;
							for (; i + 8 <= my_ml; i += 8) { uint64_t v; __builtin_memcpy(&v, ms + i, 8); __builtin_memcpy(md + i, &v, 8); }
							for (; i < my_ml; i++) md[i] = ms[i];
						} else {
							uint32_t m = 0;	/* i mod offset, carried */
							for (uint32_t i = 0; i < my_ml; i++) { md[i] = ms[m]; m = m + 1 == my_off ? 0 : m + 1; }
						}
						pendm = false;
					}
					wave_fence();
				}
			}
		}
		f->rep[0] = sd.r0; f->rep[1] = sd.r1; f->rep[2] = sd.r2;
		if (sd.pos != 0) return -1;	/* (libzstd 1.5 checks the exact end too; 1.4.8 does not) */
	} else if (left != 0) return -1;
	const size_t rest = regen - lit_pos;
	if (out - dst_pos + rest > ZBLOCK_MAX) return -1;
	if (out + rest > dst_cap) return -2;
	t_copy<W>(dst + out, f->lit + lit_pos, rest); out += rest;
	return (int64_t)(out - dst_pos);
}

/* One frame at src (zstd or skippable).  *consumed = its compressed length.  Returns decoded bytes appended at
 * dst + dst_pos, or -1 format error, -2 dst too small, -3 truncated input. */
template <bool W> __device__ __forceinline__ static int64_t zstd_frame(const uint8_t *src, size_t len, uint8_t *dst, size_t dst_pos, size_t dst_cap, size_t *consumed, zframe *fp, uint8_t *litbuf, uint32_t options)
{
	if (len < 4) return -3;
	const uint32_t magic = rd32(src);
	if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
		if (len < 8) return -3;
		const uint64_t sz = rd32(src + 4);
		if (8 + sz > len) return -3;
		*consumed = (size_t)(8 + sz);
		return 0;
	}
	if (magic != 0xFD2FB528u) return -1;
	if (len < 5) return -3;
	const int fhd = src[4];
	const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, csum = (fhd >> 2) & 1, did_flag = fhd & 3;
	if (fhd & 0x08) return -5;	/* reserved bit: "Unsupported frame parameter" */
	size_t p = 5;
	uint64_t window = 0;
	if (!single) {
		if (p >= len) return -3;
		const int wd = src[p++];
		const uint64_t base = 1ull << (10 + (wd >> 3));
		window = base + (base >> 3) * (uint64_t)(wd & 7);
	}
	const int did_len[4] = { 0, 1, 2, 4 };
	if (p + (size_t)did_len[did_flag] > len) return -3;
	uint32_t did = 0;
	for (int i = 0; i < did_len[did_flag]; i++) did |= (uint32_t)src[p + i] << (8 * i);
	p += (size_t)did_len[did_flag];
	const int fcs_len = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
	if (p + (size_t)fcs_len > len) return -3;
	uint64_t fcs = 0;
	for (int i = 0; i < fcs_len; i++) fcs |= (uint64_t)src[p + i] << (8 * i);
	if (fcs_len == 2) fcs += 256;
	p += (size_t)fcs_len;
	if (single) window = fcs;
	if (did != 0) return -7;			/* no dictionary is ever loaded by the filter: "Dictionary mismatch" */
	if (window > (1ull << 27)) return -6;	/* ZSTD_decompressStream's default window limit (2^27) */
	zframe &f = *fp;
	f.have_huf = f.have_ll = f.have_of = f.have_ml = 0;
	f.rep[0] = 1; f.rep[1] = 4; f.rep[2] = 8;
	f.lit = litbuf;
	size_t out = dst_pos;
	for (;;) {
		if (p + 3 > len) return -3;
		const uint32_t bh = src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
		p += 3;
		const int last = bh & 1, type = (bh >> 1) & 3;
		const uint32_t bsize = bh >> 3;
		if (type == 3) return -1;
		if (bsize > ZBLOCK_MAX) return -1;
		if (type == 1) {
			if (p + 1 > len) return -3;
			if (out + bsize > dst_cap) return -2;
			t_fill<W>(dst + out, src[p], bsize); out += bsize; p += 1;
		} else {
			if (p + bsize > len) return -3;
			if (type == 0) {
				if (out + bsize > dst_cap) return -2;
				t_copy<W>(dst + out, src + p, bsize); out += bsize;
			} else {
				const int64_t r = zstd_block<W>(&f, src + p, bsize, dst + dst_pos, out - dst_pos, dst_cap - dst_pos);
				if (r < 0) return r;
				out += (size_t)r;
			}
			p += bsize;
		}
		if (last) break;
	}
	if (fcs_len && (uint64_t)(out - dst_pos) != fcs) return -1;
	if (csum) {
		if (p + 4 > len) return -3;
		if (W) wave_fence();
		if (!(options & LA_ZSTD_OPT_NO_VERIFY) &&
		    (uint32_t)(W ? wave_xxh64(dst + dst_pos, out - dst_pos, 0) : dev_xxh64(dst + dst_pos, out - dst_pos, 0)) != rd32(src + p)) return -4;
		p += 4;
	}
	*consumed = p;
	return (int64_t)(out - dst_pos);
}


#define ZSTD_WS_STRIDE (144u * 1024u)	/* per lane: zframe (tables) + the literals buffer of one block */
#define ZSTD_MAX_LANES 8192u

__global__ __launch_bounds__(64) void zstd_frames_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    const la_zstd_frame *__restrict__ frames, uint32_t n, uint8_t *dst, uint64_t dst_cap, la_zstd_result *results,
    uint8_t *ws, uint32_t lanes, uint32_t options)
{
	const uint32_t w = blockIdx.x * 64u + threadIdx.x;
	if (w >= lanes)
		return;
	zframe *fp = (zframe *)(ws + (size_t)w * ZSTD_WS_STRIDE);
	uint8_t *lit = (uint8_t *)fp + 12288;
	for (uint32_t k = 0; k < sizeof(seq_tabs) / 4; k++)
		((uint32_t *)&fp->tabs)[k] = ((const uint32_t *)&SEQ_TABS)[k];
	for (uint32_t i = w; i < n; i += lanes) {
		const la_zstd_frame fr = frames[i];
		la_zstd_result r;
		r.status = LA_ST_ZSTD_CORRUPT; r.reserved = 0; r.out_len = 0;
		if (fr.src_off <= src_bytes && fr.src_len <= src_bytes - fr.src_off && fr.dst_off <= dst_cap && fr.dst_cap <= dst_cap - fr.dst_off) {
			size_t used = 0;
			const int64_t v = zstd_frame<false>(src + fr.src_off, (size_t)fr.src_len, dst + fr.dst_off, 0, (size_t)fr.dst_cap, &used, fp, lit, options);
			if (v >= 0) {
				r.status = (used == fr.src_len) ? LA_ST_OK : LA_ST_ZSTD_CORRUPT;	/* the host cut the frame here */
				r.out_len = (uint64_t)v;
			} else {
				r.status = v == -2 ? LA_ST_ZSTD_OUT_FULL : v == -3 ? LA_ST_ZSTD_TRUNCATED : v == -4 ? LA_ST_ZSTD_BAD_CHECKSUM :
				    v == -5 ? LA_ST_ZSTD_UNSUPPORTED : v == -6 ? LA_ST_ZSTD_WINDOW : v == -7 ? LA_ST_ZSTD_DICTIONARY : LA_ST_ZSTD_CORRUPT;
			}
		}
		results[i] = r;
	}
}

#define ZSTD_WAVE_WS_STRIDE (132u * 1024u)	/* per wave: the literals buffer of one block (the tables are in LDS) */
#define ZSTD_MAX_WAVES 4096u

/* one WAVE per frame: tables in LDS, uniform decode, lane-parallel byte moving, Huffman streams and XXH64 on four lanes */
__global__ __launch_bounds__(64) void zstd_frames_wave_kernel(const uint8_t *__restrict__ src, uint64_t src_bytes,
    const la_zstd_frame *__restrict__ frames, uint32_t n, uint8_t *dst, uint64_t dst_cap, la_zstd_result *results,
    uint8_t *ws, uint32_t waves, uint32_t options)
{
	__shared__ zframe sf;
	for (uint32_t i = threadIdx.x; i < sizeof(seq_tabs) / 4; i += 64)
		((uint32_t *)&sf.tabs)[i] = ((const uint32_t *)&SEQ_TABS)[i];
	__syncthreads();
	const uint32_t w = blockIdx.x;
	uint8_t *lit = ws + (size_t)w * ZSTD_WAVE_WS_STRIDE;
	for (uint32_t i = w; i < n; i += waves) {
		const la_zstd_frame fr = frames[i];
		la_zstd_result r;
		r.status = LA_ST_ZSTD_CORRUPT; r.reserved = 0; r.out_len = 0;
		if (fr.src_off <= src_bytes && fr.src_len <= src_bytes - fr.src_off && fr.dst_off <= dst_cap && fr.dst_cap <= dst_cap - fr.dst_off) {
			size_t used = 0;
			const int64_t v = zstd_frame<true>(src + fr.src_off, (size_t)fr.src_len, dst + fr.dst_off, 0, (size_t)fr.dst_cap, &used, &sf, lit, options);
			if (v >= 0) {
				r.status = (used == fr.src_len) ? LA_ST_OK : LA_ST_ZSTD_CORRUPT;
				r.out_len = (uint64_t)v;
			} else {
				r.status = v == -2 ? LA_ST_ZSTD_OUT_FULL : v == -3 ? LA_ST_ZSTD_TRUNCATED : v == -4 ? LA_ST_ZSTD_BAD_CHECKSUM :
				    v == -5 ? LA_ST_ZSTD_UNSUPPORTED : v == -6 ? LA_ST_ZSTD_WINDOW : v == -7 ? LA_ST_ZSTD_DICTIONARY : LA_ST_ZSTD_CORRUPT;
			}
		}
		if (threadIdx.x == 0)
			results[i] = r;
		wave_fence();
	}
}

static uint32_t zstd_lanes(uint32_t n) { return n < ZSTD_MAX_LANES ? n : ZSTD_MAX_LANES; }
static uint32_t zstd_waves(uint32_t n) { return n < ZSTD_MAX_WAVES ? n : ZSTD_MAX_WAVES; }

uint64_t la_zstd_workspace_bytes(uint32_t n_frames)
{
	const uint64_t a = (uint64_t)zstd_lanes(n_frames) * ZSTD_WS_STRIDE, b = (uint64_t)zstd_waves(n_frames) * ZSTD_WAVE_WS_STRIDE;
	return a > b ? a : b;
}

void la_launch_zstd_frames(hipStream_t s, const uint8_t *d_src, uint64_t src_bytes, const la_zstd_frame *d_frames, uint32_t n,
    uint8_t *d_dst, uint64_t dst_cap, la_zstd_result *d_results, uint8_t *ws, uint32_t options)
{
	if (n == 0) return;
	static_assert(sizeof(zframe) <= 12288, "zframe must fit in front of the literals buffer");
	if (options & LA_ZSTD_OPT_LANE_KERNEL) {
		const uint32_t lanes = zstd_lanes(n);
		hipLaunchKernelGGL(zstd_frames_kernel, dim3((lanes + 63u) / 64u), dim3(64), 0, s, d_src, src_bytes, d_frames, n, d_dst, dst_cap,
		    d_results, ws, lanes, options);
	} else {
		const uint32_t waves = zstd_waves(n);
		hipLaunchKernelGGL(zstd_frames_wave_kernel, dim3(waves), dim3(64), 0, s, d_src, src_bytes, d_frames, n, d_dst, dst_cap,
		    d_results, ws, waves, options);
	}
}
