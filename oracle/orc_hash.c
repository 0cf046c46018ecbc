/*
 * orc_hash.c -- ORACLE (test infrastructure, see la_oracle.h): XXH32 and CRC32.
 *
 * XXH32 follows the xxHash-32 specification as implemented by the reference's
 * libarchive/xxhash.c (constants :189-193, one-shot :234-291, streaming
 * :347-507).  CRC32 follows libarchive/archive_crc32.h:43-84 (reflected
 * polynomial 0xEDB88320, pre/post inversion).
 */
#include "la_oracle.h"
#include <string.h>

#define P1 0x9E3779B1u
#define P2 0x85EBCA77u
#define P3 0xC2B2AE3Du
#define P4 0x27D4EB2Fu
#define P5 0x165667B1u

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t rd32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint32_t xxh_round(uint32_t acc, uint32_t lane)
{
	return rotl32(acc + lane * P2, 13) * P1;
}

static uint32_t xxh_finish(uint32_t h, const uint8_t *p, size_t rem)
{
	while (rem >= 4) {
		h = rotl32(h + rd32(p) * P3, 17) * P4;
		p += 4; rem -= 4;
	}
	while (rem > 0) {
		h = rotl32(h + (uint32_t)(*p) * P5, 11) * P1;
		p++; rem--;
	}
	h ^= h >> 15; h *= P2;
	h ^= h >> 13; h *= P3;
	h ^= h >> 16;
	return h;
}

/* xxhash.c:234-291: `len` is an unsigned int there, so the length added to the
 * hash is taken modulo 2^32. */
uint32_t orc_xxh32(const void *input, size_t len, uint32_t seed)
{
	const uint8_t *p = (const uint8_t *)input;
	const uint8_t *end = p + len;
	uint32_t h;

	if (len >= 16) {
		uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
		const uint8_t *limit = end - 16;
		do {
			v1 = xxh_round(v1, rd32(p));
			v2 = xxh_round(v2, rd32(p + 4));
			v3 = xxh_round(v3, rd32(p + 8));
			v4 = xxh_round(v4, rd32(p + 12));
			p += 16;
		} while (p <= limit);
		h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
	} else {
		h = seed + P5;
	}
	h += (uint32_t)len;
	return xxh_finish(h, p, (size_t)(end - p));
}

void orc_xxh32_init(orc_xxh32_state *st, uint32_t seed)
{
	memset(st, 0, sizeof(*st));
	st->seed = seed;
	st->v[0] = seed + P1 + P2;
	st->v[1] = seed + P2;
	st->v[2] = seed;
	st->v[3] = seed - P1;
}

void orc_xxh32_update(orc_xxh32_state *st, const void *input, size_t len)
{
	const uint8_t *p = (const uint8_t *)input;
	const uint8_t *end = p + len;

	st->total_len += len;
	if (st->memsize + len < 16) {
		memcpy(st->mem + st->memsize, p, len);
		st->memsize += (uint32_t)len;
		return;
	}
	if (st->memsize) {
		size_t fill = 16 - st->memsize;
		memcpy(st->mem + st->memsize, p, fill);
		for (int i = 0; i < 4; i++)
			st->v[i] = xxh_round(st->v[i], rd32(st->mem + 4 * i));
		p += fill;
		st->memsize = 0;
	}
	while (p + 16 <= end) {
		for (int i = 0; i < 4; i++)
			st->v[i] = xxh_round(st->v[i], rd32(p + 4 * i));
		p += 16;
	}
	if (p < end) {
		memcpy(st->mem, p, (size_t)(end - p));
		st->memsize = (uint32_t)(end - p);
	}
}

uint32_t orc_xxh32_digest(const orc_xxh32_state *st)
{
	uint32_t h;
	if (st->total_len >= 16)
		h = rotl32(st->v[0], 1) + rotl32(st->v[1], 7) + rotl32(st->v[2], 12) + rotl32(st->v[3], 18);
	else
		h = st->seed + P5;
	h += (uint32_t)st->total_len;
	return xxh_finish(h, st->mem, st->memsize);
}

/* ---------------- CRC32 ---------------- */

static uint32_t crc_tab[256];
static int crc_tab_ready;

static void crc_build(void)
{
	for (uint32_t b = 0; b < 256; b++) {
		uint32_t c = b;
		for (int k = 0; k < 8; k++)
			c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : (c >> 1);
		crc_tab[b] = c;
	}
	crc_tab_ready = 1;
}

/* archive_crc32.h:43-84.  A NULL buffer returns 0 (the "initial value" call). */
uint32_t orc_crc32(uint32_t crc, const void *buf, size_t len)
{
	const uint8_t *p = (const uint8_t *)buf;
	if (p == NULL)
		return 0;
	if (!crc_tab_ready)
		crc_build();
	crc = ~crc;
	while (len--)
		crc = crc_tab[(crc ^ *p++) & 0xff] ^ (crc >> 8);
	return ~crc;
}

/* multiply a(x)*b(x) mod P(x) in the reflected representation */
static uint32_t gf2_mulmod(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
	for (int i = 0; i < 32; i++) {
		if (a & 0x80000000u)
			r ^= b;
		a <<= 1;
		b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : (b >> 1);
	}
	return r;
}

/* x^(8*n) mod P, reflected */
static uint32_t gf2_xpow8n(uint64_t n)
{
	uint32_t result = 0x80000000u;	/* the polynomial "1" */
	uint32_t base = 0x00800000u;	/* x^8 */
	while (n) {
		if (n & 1)
			result = gf2_mulmod(result, base);
		base = gf2_mulmod(base, base);
		n >>= 1;
	}
	return result;
}

uint32_t orc_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b)
{
	return gf2_mulmod(crc_a, gf2_xpow8n(len_b)) ^ crc_b;
}
