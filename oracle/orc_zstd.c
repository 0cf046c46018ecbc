/*
 * orc_zstd.c -- CPU restatement of the Zstandard frame format (RFC 8878) for the zstd read filter.
 *
 * TEST INFRASTRUCTURE ONLY (checker and CPU baseline): nothing under libarchive_amd/ links or calls it.
 *
 * The reference's filter (libarchive/archive_read_support_filter_zstd.c:171-260) hands the stream to libzstd's
 * ZSTD_decompressStream(); libzstd is a third-party dependency that is not part of the reference tree (the image
 * carries libzstd.so.1 1.4.8 without headers).  This file restates the published format; tests/test_oracle_zstd.py
 * pins it against (i) the reference's own zstd fixtures (cat/test/test_expand.zst.uu, test_empty.zst.uu,
 * libarchive/test/test_compat_zstd_{1,2}.tar.zst.uu) and (ii) the image's libzstd on randomized inputs at every
 * compression level (ZSTD_compress -> this decoder, and ZSTD_decompress verdicts on damaged streams).
 *
 * Stream semantics follow the filter: frames back to back, skippable frames (0x184D2A5x) skipped
 * (zstd.c:117-130 bids on both magics), end of input inside a frame = "Truncated zstd input" (zstd.c:213-217),
 * any format error = "Zstd decompression failed" (zstd.c:226-231).
 */
#include "la_oracle.h"
#include <string.h>
#include <stdlib.h>

/* ---- XXH64 (content checksum = low 32 bits, RFC 8878 3.1.1) ---- */
#define P64_1 11400714785074694791ULL
#define P64_2 14029467366897019727ULL
#define P64_3 1609587929392839161ULL
#define P64_4 9650029242287828579ULL
#define P64_5 2870177450012600261ULL
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t xxh64_round(uint64_t acc, uint64_t in) { acc += in * P64_2; acc = rotl64(acc, 31); return acc * P64_1; }
static uint64_t xxh64_merge(uint64_t h, uint64_t v) { v = xxh64_round(0, v); h ^= v; return h * P64_1 + P64_4; }

uint64_t orc_xxh64(const void *input, size_t len, uint64_t seed)
{
	const uint8_t *p = (const uint8_t *)input, *end = p + len;
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

/* line of the check that refused the stream last (diagnostics of the tests) */
int orc_zstd_last_line;
static int zfail(int line) { orc_zstd_last_line = line; return -1; }

/* ---- bit readers ---- */
/* n (<= 32) bits at bit position pos of the little-endian bit array src[0..len); positions outside read as zero */
static uint32_t bits_at(const uint8_t *src, size_t len, int64_t pos, unsigned n)
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
static int highbit(uint32_t v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

/* backward stream: returns the bit position just below the end marker, or -1 when the last byte is zero */
static int64_t rev_init(const uint8_t *src, size_t len)
{
	if (len == 0 || src[len - 1] == 0) return zfail(__LINE__);
	return (int64_t)(len - 1) * 8 + highbit(src[len - 1]);
}
static uint32_t rev_read(const uint8_t *src, size_t len, int64_t *pos, unsigned n)
{
	*pos -= n;
	return bits_at(src, len, *pos, n);
}

/* ---- FSE ---- */
typedef struct { uint8_t sym, nbits; uint16_t base; } fse_ent;
typedef struct { fse_ent e[512]; int al; } fse_tab;

/* normalized counts (RFC 8878 4.1.1); returns bytes consumed or -1 */
static int fse_read_ncount(const uint8_t *src, size_t len, int max_al, int max_sym, int16_t *norm, int *n_sym, int *al_out)
{
	int64_t bp = 0;
	if (len == 0) return zfail(__LINE__);
	const int al = (int)bits_at(src, len, bp, 4) + 5; bp += 4;
	if (al > max_al) return zfail(__LINE__);
	int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0;
	while (remaining > 1 && sym <= max_sym) {
		if ((size_t)((bp + 7) >> 3) > len + 4) return zfail(__LINE__);
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
				for (uint32_t i = 0; i < r; i++) { if (sym > max_sym) return zfail(__LINE__); norm[sym++] = 0; }
				if (r != 3) break;
			}
		}
		if (remaining < 1) return zfail(__LINE__);
		while (remaining < threshold) { nbits--; threshold >>= 1; }
	}
	if (remaining != 1 || sym > max_sym + 1) return zfail(__LINE__);
	const size_t used = (size_t)((bp + 7) >> 3);
	if (used > len) return zfail(__LINE__);
	*n_sym = sym; *al_out = al;
	return (int)used;
}

static int fse_build(fse_tab *t, const int16_t *norm, int n_sym, int al)
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
	if (pos != 0) return zfail(__LINE__);
	for (int u = 0; u < size; u++) {
		const int s = t->e[u].sym;
		const int nx = next[s]++;
		const int nb = al - highbit((uint32_t)nx);
		t->e[u].nbits = (uint8_t)nb;
		t->e[u].base = (uint16_t)((nx << nb) - size);
	}
	return 0;
}
static void fse_rle(fse_tab *t, int sym) { t->al = 0; t->e[0].sym = (uint8_t)sym; t->e[0].nbits = 0; t->e[0].base = 0; }

/* ---- Huffman (RFC 8878 4.2) ---- */
typedef struct { uint8_t sym[2048], nbits[2048]; int maxbits; } huf_tab;

static int huf_read(huf_tab *h, const uint8_t *src, size_t len)	/* returns bytes consumed or -1 */
{
	uint8_t w[256];
	int n = 0;
	size_t used;
	if (len < 1) return zfail(__LINE__);
	const int hb = src[0];
	if (hb >= 128) {	/* direct: 4-bit weights */
		n = hb - 127;
		used = 1 + (size_t)(n + 1) / 2;
		if (used > len) return zfail(__LINE__);
		for (int i = 0; i < n; i++)
			w[i] = (i & 1) ? (src[1 + i / 2] & 15) : (src[1 + i / 2] >> 4);
	} else {		/* FSE-compressed weights, two interleaved states */
		used = 1 + (size_t)hb;
		if (hb == 0 || used > len) return zfail(__LINE__);
		int16_t norm[16]; int ns, al;
		fse_tab t;
		const int c = fse_read_ncount(src + 1, (size_t)hb, 6, 11, norm, &ns, &al);
		if (c < 0 || fse_build(&t, norm, ns, al) < 0) return zfail(__LINE__);
		const uint8_t *bs = src + 1 + c; const size_t bl = (size_t)hb - (size_t)c;
		int64_t pos = rev_init(bs, bl);
		if (pos < 0) return zfail(__LINE__);
		uint32_t s1 = rev_read(bs, bl, &pos, (unsigned)al), s2 = rev_read(bs, bl, &pos, (unsigned)al);
		if (pos < 0) return zfail(__LINE__);
		for (;;) {
			if (n > 253) return zfail(__LINE__);
			w[n++] = t.e[s1].sym;
			s1 = t.e[s1].base + rev_read(bs, bl, &pos, t.e[s1].nbits);
			if (pos < 0) { w[n++] = t.e[s2].sym; break; }
			if (n > 253) return zfail(__LINE__);
			w[n++] = t.e[s2].sym;
			s2 = t.e[s2].base + rev_read(bs, bl, &pos, t.e[s2].nbits);
			if (pos < 0) { w[n++] = t.e[s1].sym; break; }
		}
	}
	/* the last weight completes a power of two */
	uint32_t sum = 0;
	for (int i = 0; i < n; i++) { if (w[i] > 11) return zfail(__LINE__); if (w[i]) sum += 1u << (w[i] - 1); }
	if (sum == 0) return zfail(__LINE__);
	const int maxbits = highbit(sum) + 1;
	if (maxbits > 11) return zfail(__LINE__);
	const uint32_t left = (1u << maxbits) - sum;
	if (left == 0 || (left & (left - 1))) return zfail(__LINE__);
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
	if (pos != (1u << maxbits)) return zfail(__LINE__);
	return (int)used;
}

static int huf_stream(const huf_tab *h, const uint8_t *src, size_t len, uint8_t *out, size_t n)
{
	int64_t pos = rev_init(src, len);
	if (pos < 0) return zfail(__LINE__);
	for (size_t i = 0; i < n; i++) {
		const uint32_t idx = bits_at(src, len, pos - h->maxbits, (unsigned)h->maxbits);
		out[i] = h->sym[idx];
		pos -= h->nbits[idx];
		if (pos < 0) return zfail(__LINE__);
	}
	return pos == 0 ? 0 : -1;
}

/* ---- sequences ---- */
static const uint32_t LL_BASE[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536 };
static const uint8_t  LL_BITS[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const uint32_t ML_BASE[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539 };
static const uint8_t  ML_BITS[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static const int16_t LL_DEF[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static const int16_t ML_DEF[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static const int16_t OF_DEF[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

typedef struct {
	huf_tab huf; int have_huf;
	fse_tab ll, of, ml; int have_ll, have_of, have_ml;
	uint32_t rep[3];
	uint8_t *lit;	/* 128 KiB + slack */
} zframe;

/* one table of the sequences section; returns bytes consumed or -1 */
static int seq_table(fse_tab *t, int *have, int mode, const uint8_t *src, size_t len, int max_al, int max_sym,
    const int16_t *def, int def_n, int def_al)
{
	if (mode == 0) { if (fse_build(t, def, def_n, def_al) < 0) return zfail(__LINE__); *have = 1; return 0; }
	if (mode == 1) { if (len < 1 || src[0] > max_sym) return zfail(__LINE__); fse_rle(t, src[0]); *have = 1; return 1; }
	if (mode == 2) {
		int16_t norm[64]; int ns, al;
		const int c = fse_read_ncount(src, len, max_al, max_sym, norm, &ns, &al);
		if (c < 0 || fse_build(t, norm, ns, al) < 0) return zfail(__LINE__);
		*have = 1;
		return c;
	}
	return *have ? 0 : -1;	/* repeat */
}

#define ZBLOCK_MAX (128u * 1024u)

/* one compressed block; returns bytes produced or -1 */
static int64_t zstd_block(zframe *f, const uint8_t *src, size_t len, uint8_t *dst, size_t dst_pos, size_t dst_cap)
{
	if (len < 1) return zfail(__LINE__);	/* (libzstd: a compressed block needs at least a literals header) */
	/* ---- literals section ---- */
	const int ltype = src[0] & 3, sf = (src[0] >> 2) & 3;
	size_t hl, regen, comp = 0;
	int streams = 1;
	if (ltype < 2) {
		if (sf == 0 || sf == 2) { hl = 1; regen = src[0] >> 3; }
		else if (sf == 1) { if (len < 2) return zfail(__LINE__); hl = 2; regen = (src[0] >> 4) | ((size_t)src[1] << 4); }
		else { if (len < 3) return zfail(__LINE__); hl = 3; regen = (src[0] >> 4) | ((size_t)src[1] << 4) | ((size_t)src[2] << 12); }
	} else {
		if (sf < 2) {
			if (len < 3) return zfail(__LINE__);
			hl = 3; streams = sf == 0 ? 1 : 4;
			const uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16);
			regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF;
		} else if (sf == 2) {
			if (len < 4) return zfail(__LINE__);
			hl = 4; streams = 4;
			const uint32_t v = rd32(src);
			regen = (v >> 4) & 0x3FFF; comp = v >> 18;
		} else {
			if (len < 5) return zfail(__LINE__);
			hl = 5; streams = 4;
			const uint64_t v = (uint64_t)rd32(src) | ((uint64_t)src[4] << 32);
			regen = (size_t)((v >> 4) & 0x3FFFF); comp = (size_t)(v >> 22);
		}
	}
	if (regen > ZBLOCK_MAX) return zfail(__LINE__);
	const uint8_t *p = src + hl;
	size_t left = len - hl;
	if (ltype == 0) { if (regen > left) return zfail(__LINE__); memcpy(f->lit, p, regen); p += regen; left -= regen; }
	else if (ltype == 1) { if (left < 1) return zfail(__LINE__); memset(f->lit, p[0], regen); p += 1; left -= 1; }
	else {
		if (comp > left) return zfail(__LINE__);
		const uint8_t *hp = p; size_t hleft = comp;
		if (ltype == 2) {
			const int c = huf_read(&f->huf, hp, hleft);
			if (c < 0) return zfail(__LINE__);
			f->have_huf = 1; hp += c; hleft -= (size_t)c;
		} else if (!f->have_huf) return zfail(__LINE__);
		if (streams == 1) {
			if (huf_stream(&f->huf, hp, hleft, f->lit, regen) < 0) return zfail(__LINE__);
		} else {
			if (hleft < 6) return zfail(__LINE__);
			const size_t s1 = hp[0] | ((size_t)hp[1] << 8), s2 = hp[2] | ((size_t)hp[3] << 8), s3 = hp[4] | ((size_t)hp[5] << 8);
			if (6 + s1 + s2 + s3 > hleft) return zfail(__LINE__);
			const size_t s4 = hleft - 6 - s1 - s2 - s3, q = (regen + 3) / 4;
			if (3 * q > regen) return zfail(__LINE__);
			hp += 6;
			if (huf_stream(&f->huf, hp, s1, f->lit, q) < 0) return zfail(__LINE__);
			if (huf_stream(&f->huf, hp + s1, s2, f->lit + q, q) < 0) return zfail(__LINE__);
			if (huf_stream(&f->huf, hp + s1 + s2, s3, f->lit + 2 * q, q) < 0) return zfail(__LINE__);
			if (huf_stream(&f->huf, hp + s1 + s2 + s3, s4, f->lit + 3 * q, regen - 3 * q) < 0) return zfail(__LINE__);
		}
		p += comp; left -= comp;
	}
	/* ---- sequences section ---- */
	if (left < 1) return zfail(__LINE__);
	size_t nseq = p[0];
	if (nseq == 0) { p += 1; left -= 1; }
	else if (nseq < 128) { p += 1; left -= 1; }
	else if (nseq < 255) { if (left < 2) return zfail(__LINE__); nseq = ((nseq - 128) << 8) + p[1]; p += 2; left -= 2; }
	else { if (left < 3) return zfail(__LINE__); nseq = p[1] + ((size_t)p[2] << 8) + 0x7F00; p += 3; left -= 3; }
	size_t out = dst_pos, lit_pos = 0;
	if (nseq) {
		if (left < 1) return zfail(__LINE__);
		const int modes = p[0];
		/* (bits 0-1 are reserved; libzstd 1.4.8 ZSTD_decodeSeqHeaders does not look at them) */
		p += 1; left -= 1;
		int c;
		c = seq_table(&f->ll, &f->have_ll, modes >> 6, p, left, 9, 35, LL_DEF, 36, 6); if (c < 0) return zfail(__LINE__); p += c; left -= (size_t)c;
		c = seq_table(&f->of, &f->have_of, (modes >> 4) & 3, p, left, 8, 31, OF_DEF, 29, 5); if (c < 0) return zfail(__LINE__); p += c; left -= (size_t)c;
		c = seq_table(&f->ml, &f->have_ml, (modes >> 2) & 3, p, left, 9, 52, ML_DEF, 53, 6); if (c < 0) return zfail(__LINE__); p += c; left -= (size_t)c;
		int64_t pos = rev_init(p, left);
		if (pos < 0) return zfail(__LINE__);
		uint32_t sl = rev_read(p, left, &pos, (unsigned)f->ll.al);
		uint32_t so = rev_read(p, left, &pos, (unsigned)f->of.al);
		uint32_t sm = rev_read(p, left, &pos, (unsigned)f->ml.al);
		if (pos < 0) return zfail(__LINE__);
		for (size_t i = 0; i < nseq; i++) {
			const int oc = f->of.e[so].sym, mc = f->ml.e[sm].sym, lc = f->ll.e[sl].sym;
			if (oc > 31 || mc > 52 || lc > 35) return zfail(__LINE__);
			const uint32_t ov = (oc ? ((1u << oc) + rev_read(p, left, &pos, (unsigned)oc)) : 1u);
			const uint32_t ml = ML_BASE[mc] + rev_read(p, left, &pos, ML_BITS[mc]);
			const uint32_t ll = LL_BASE[lc] + rev_read(p, left, &pos, LL_BITS[lc]);
			if (pos < 0) return zfail(__LINE__);
			uint32_t offset;
			if (ov > 3) {
				offset = ov - 3;
				f->rep[2] = f->rep[1]; f->rep[1] = f->rep[0]; f->rep[0] = offset;
			} else {
				uint32_t idx = ov - 1 + (ll == 0 ? 1u : 0u);	/* 0..3 */
				if (idx == 0) offset = f->rep[0];
				else {
					offset = idx == 3 ? f->rep[0] - 1 : f->rep[idx];
					if (offset == 0) return zfail(__LINE__);
					if (idx != 1) f->rep[2] = f->rep[1];
					f->rep[1] = f->rep[0]; f->rep[0] = offset;
				}
			}
			if (i + 1 < nseq) {
				sl = f->ll.e[sl].base + rev_read(p, left, &pos, f->ll.e[sl].nbits);
				sm = f->ml.e[sm].base + rev_read(p, left, &pos, f->ml.e[sm].nbits);
				so = f->of.e[so].base + rev_read(p, left, &pos, f->of.e[so].nbits);
				if (pos < 0) return zfail(__LINE__);
			}
			/* execute */
			if (ll > regen - lit_pos) return zfail(__LINE__);
			if (out - dst_pos + ll + ml > ZBLOCK_MAX) return zfail(__LINE__);
			if (out + ll + ml > dst_cap) return -2;
			memcpy(dst + out, f->lit + lit_pos, ll); out += ll; lit_pos += ll;
			if (offset > out) return zfail(__LINE__);
			for (uint32_t k = 0; k < ml; k++) dst[out + k] = dst[out + k - offset];
			out += ml;
		}
		if (pos != 0) return zfail(__LINE__);	/* (libzstd 1.5 checks the exact end too; 1.4.8 does not) */
	} else if (left != 0) return zfail(__LINE__);
	const size_t rest = regen - lit_pos;
	if (out - dst_pos + rest > ZBLOCK_MAX) return zfail(__LINE__);
	if (out + rest > dst_cap) return -2;
	memcpy(dst + out, f->lit + lit_pos, rest); out += rest;
	return (int64_t)(out - dst_pos);
}

/* One frame at src (zstd or skippable).  *consumed = its compressed length.  Returns decoded bytes appended at
 * dst + dst_pos, or -1 format error, -2 dst too small, -3 truncated input. */
static int64_t zstd_frame(const uint8_t *src, size_t len, uint8_t *dst, size_t dst_pos, size_t dst_cap, size_t *consumed, uint8_t *litbuf)
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
	if (magic != 0xFD2FB528u) return zfail(__LINE__);
	if (len < 5) return -3;
	const int fhd = src[4];
	const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, csum = (fhd >> 2) & 1, did_flag = fhd & 3;
	if (fhd & 0x08) return zfail(__LINE__);	/* reserved bit */
	size_t p = 5;
	uint64_t window = 0;
	if (!single) {
		if (p >= len) return -3;
		const int wd = src[p++];
		const uint64_t base = 1ull << (10 + (wd >> 3));
		window = base + (base >> 3) * (uint64_t)(wd & 7);
	}
	static const int did_len[4] = { 0, 1, 2, 4 };
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
	if (did != 0) return zfail(__LINE__);			/* no dictionary is ever loaded by the filter */
	if (window > (1ull << 27)) return zfail(__LINE__);	/* ZSTD_decompressStream's default window limit (2^27) */
	zframe f;
	memset(&f, 0, sizeof(f));
	f.rep[0] = 1; f.rep[1] = 4; f.rep[2] = 8;
	f.lit = litbuf;
	size_t out = dst_pos;
	for (;;) {
		if (p + 3 > len) return -3;
		const uint32_t bh = src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
		p += 3;
		const int last = bh & 1, type = (bh >> 1) & 3;
		const uint32_t bsize = bh >> 3;
		if (type == 3) return zfail(__LINE__);
		if (bsize > ZBLOCK_MAX) return zfail(__LINE__);
		if (type == 1) {
			if (p + 1 > len) return -3;
			if (out + bsize > dst_cap) return -2;
			memset(dst + out, src[p], bsize); out += bsize; p += 1;
		} else {
			if (p + bsize > len) return -3;
			if (type == 0) {
				if (out + bsize > dst_cap) return -2;
				memcpy(dst + out, src + p, bsize); out += bsize;
			} else {
				const int64_t r = zstd_block(&f, src + p, bsize, dst + dst_pos, out - dst_pos, dst_cap - dst_pos);
				if (r < 0) return r;
				out += (size_t)r;
			}
			p += bsize;
		}
		if (last) break;
	}
	if (fcs_len && (uint64_t)(out - dst_pos) != fcs) return zfail(__LINE__);
	if (csum) {
		if (p + 4 > len) return -3;
		if ((uint32_t)orc_xxh64(dst + dst_pos, out - dst_pos, 0) != rd32(src + p)) return zfail(__LINE__);
		p += 4;
	}
	*consumed = p;
	return (int64_t)(out - dst_pos);
}

int orc_zstd_bid(const uint8_t *p, size_t avail)
{
	/* libarchive/archive_read_support_filter_zstd.c:107-131 */
	if (avail < 4) return 0;
	const uint32_t m = rd32(p);
	if (m == 0xFD2FB528u) return 32;
	if ((m & 0xFFFFFFF0u) == 0x184D2A50u) return 32;
	return 0;
}

/* Whole stream through the filter's semantics.  Returns ORC_OK / ORC_FATAL like the other stream decoders;
 * *out_len = bytes produced before the verdict; msg = the filter's error string ("" on success). */
int orc_zstd_stream_decode(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *out_len, char *msg, size_t msg_cap)
{
	size_t p = 0, out = 0;
	uint8_t *lit = (uint8_t *)malloc(ZBLOCK_MAX + 64);
	if (msg && msg_cap) msg[0] = 0;
	int rc = 0;
	while (p < src_len) {
		size_t used = 0;
		const int64_t r = zstd_frame(src + p, src_len - p, dst, out, dst_cap, &used, lit);
		if (r < 0) {
			if (msg && msg_cap) {
				const char *m = r == -3 ? "Truncated zstd input" : (r == -2 ? "oracle: output buffer too small" : "Zstd decompression failed");
				strncpy(msg, m, msg_cap - 1); msg[msg_cap - 1] = 0;
			}
			rc = r == -2 ? -2 : -30;	/* ARCHIVE_FATAL */
			break;
		}
		out += (size_t)r;
		p += used;
	}
	free(lit);
	*out_len = out;
	return rc;
}
