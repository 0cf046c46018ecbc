/*
 * la_hash_dropin.c -- host-callable drop-ins for the two hash interfaces the read (and the
 * lz4 write) filters reach through libarchive's internal tables:
 *
 *   __archive_xxhash     libarchive/xxhash.c:509-515, type archive_xxhash.h:37-44
 *                        { XXH32, XXH32_init, XXH32_update, XXH32_digest }; the state comes from
 *                        malloc() and is free()-able (lz4.c:733 frees it directly), digest
 *                        frees it (xxhash.c:504)
 *   crc32()              libarchive/archive_crc32.h:43-84: zlib's signature,
 *                        crc32(x, NULL, n) == 0
 *
 * Inside libarchive (-DLA_IN_LIBARCHIVE) the table is exported under the reference's own name
 * and replaces xxhash.c; outside it is `la_archive_xxhash`.
 *
 * Where the arithmetic runs.  The device forms of these hashes are the BATCH entry points of
 * include/la_gpu.h (la_gpu_xxh32_many / la_gpu_crc32_many and the checksums fused into the decode
 * kernels): that is what the filters use for payload bytes.  The table below exists for callers
 * that hand over ONE buffer at a time:
 *   - XXH32 of one buffer is one serial multiply chain (four accumulators, no parallel form);
 *     a single chain runs faster on a host core (~2.7 GB/s) than on one GPU wave (~1 GB/s,
 *     DESIGN.md section 6), so one-shot and streaming calls are computed here, on the calling thread.
 *     Typical callers: the 2..14 descriptor bytes of a frame header, the write filter's per-block sums.
 *   - crc32 of one buffer IS parallel (GF(2) combine).  Buffers of LA_HASH_GPU_MIN bytes or more are
 *     cut into 64 KiB ranges, reduced by la_gpu_crc32_many on the device and combined here;
 *     smaller ones (and any call when no device can be opened) take the table-driven loop below.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/la_gpu.h"
#include "../../include/la_host.h"

/* ------------------------------------------------------------------ XXH32 */

#define P1 0x9E3779B1u
#define P2 0x85EBCA77u
#define P3 0xC2B2AE3Du
#define P4 0x27D4EB2Fu
#define P5 0x165667B1u

static uint32_t rd32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t round1(uint32_t v, uint32_t x) { return rotl(v + x * P2, 13) * P1; }

/* tail + avalanche shared by the one-shot and the streaming form (xxhash.c:268-291, :471-497) */
static uint32_t finish(uint32_t h, const uint8_t *p, const uint8_t *end)
{
	while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
	while (p < end) { h = rotl(h + (uint32_t)(*p) * P5, 11) * P1; p++; }
	h ^= h >> 15; h *= P2;
	h ^= h >> 13; h *= P3;
	h ^= h >> 16;
	return h;
}

/* the state layout follows the reference's XXH_state32_t (xxhash.c:325-335): 48 bytes, malloc'ed */
struct la_xxh_state {
	uint64_t total_len;
	uint32_t seed, v1, v2, v3, v4;
	int memsize;
	char memory[16];
};

static unsigned int la_XXH32(const void *input, unsigned int len, unsigned int seed)
{
	const uint8_t *p = (const uint8_t *)input, *end = p + len;
	uint32_t h;
	if (len >= 16) {
		uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
		const uint8_t *limit = end - 16;
		do {
			v1 = round1(v1, rd32(p)); v2 = round1(v2, rd32(p + 4));
			v3 = round1(v3, rd32(p + 8)); v4 = round1(v4, rd32(p + 12));
			p += 16;
		} while (p <= limit);
		h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
	} else {
		h = seed + P5;
	}
	return finish(h + len, p, end);
}

static void *la_XXH32_init(unsigned int seed)
{
	struct la_xxh_state *s = (struct la_xxh_state *)malloc(sizeof(*s));
	if (!s)
		return NULL;
	s->seed = seed;
	s->v1 = seed + P1 + P2; s->v2 = seed + P2; s->v3 = seed; s->v4 = seed - P1;
	s->total_len = 0;
	s->memsize = 0;
	memset(s->memory, 0, sizeof(s->memory));
	return s;
}

/* 0 = XXH_OK, 1 = XXH_ERROR (archive_xxhash.h:35) */
static int la_XXH32_update(void *state, const void *input, unsigned int len)
{
	struct la_xxh_state *s = (struct la_xxh_state *)state;
	const uint8_t *p = (const uint8_t *)input, *end = p + len;
	if (!input)
		return 1;
	s->total_len += len;
	if (s->memsize + len < 16) {		/* not a whole stripe yet */
		memcpy(s->memory + s->memsize, p, len);
		s->memsize += (int)len;
		return 0;
	}
	if (s->memsize) {			/* finish the carried stripe */
		memcpy(s->memory + s->memsize, p, (size_t)(16 - s->memsize));
		const uint8_t *m = (const uint8_t *)s->memory;
		s->v1 = round1(s->v1, rd32(m)); s->v2 = round1(s->v2, rd32(m + 4));
		s->v3 = round1(s->v3, rd32(m + 8)); s->v4 = round1(s->v4, rd32(m + 12));
		p += 16 - s->memsize;
		s->memsize = 0;
	}
	if (p + 16 <= end) {
		uint32_t v1 = s->v1, v2 = s->v2, v3 = s->v3, v4 = s->v4;
		const uint8_t *limit = end - 16;
		do {
			v1 = round1(v1, rd32(p)); v2 = round1(v2, rd32(p + 4));
			v3 = round1(v3, rd32(p + 8)); v4 = round1(v4, rd32(p + 12));
			p += 16;
		} while (p <= limit);
		s->v1 = v1; s->v2 = v2; s->v3 = v3; s->v4 = v4;
	}
	if (p < end) {
		memcpy(s->memory, p, (size_t)(end - p));
		s->memsize = (int)(end - p);
	}
	return 0;
}

static unsigned int la_XXH32_digest(void *state)
{
	struct la_xxh_state *s = (struct la_xxh_state *)state;
	uint32_t h;
	if (s->total_len >= 16)
		h = rotl(s->v1, 1) + rotl(s->v2, 7) + rotl(s->v3, 12) + rotl(s->v4, 18);
	else
		h = s->seed + P5;
	h += (uint32_t)s->total_len;	/* the reference adds the low 32 bits only (xxhash.c:468) */
	h = finish(h, (const uint8_t *)s->memory, (const uint8_t *)s->memory + s->memsize);
	free(s);			/* digest releases the state (xxhash.c:504) */
	return h;
}

#ifdef LA_IN_LIBARCHIVE
const struct la_archive_xxhash __archive_xxhash = {
#else
const struct la_archive_xxhash la_archive_xxhash = {
#endif
	la_XXH32, la_XXH32_init, la_XXH32_update, la_XXH32_digest
};

/* ------------------------------------------------------------------ CRC32 */

static uint32_t crc_tab[8][256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;

static void crc_build(void)
{
	for (uint32_t b = 0; b < 256; b++) {
		uint32_t c = b;
		for (int i = 0; i < 8; i++)
			c = (c >> 1) ^ ((c & 1) ? 0xEDB88320u : 0);
		crc_tab[0][b] = c;
	}
	for (uint32_t b = 0; b < 256; b++)
		for (int k = 1; k < 8; k++)
			crc_tab[k][b] = (crc_tab[k - 1][b] >> 8) ^ crc_tab[0][crc_tab[k - 1][b] & 0xff];
}

static uint32_t crc_host(uint32_t crc, const uint8_t *p, size_t len)
{
	pthread_once(&crc_once, crc_build);
	crc = ~crc;
	while (len && ((uintptr_t)p & 7)) { crc = (crc >> 8) ^ crc_tab[0][(crc ^ *p++) & 0xff]; len--; }
	while (len >= 8) {
		uint32_t a = crc ^ rd32(p), b = rd32(p + 4);
		crc = crc_tab[7][a & 0xff] ^ crc_tab[6][(a >> 8) & 0xff] ^ crc_tab[5][(a >> 16) & 0xff] ^ crc_tab[4][a >> 24] ^
		    crc_tab[3][b & 0xff] ^ crc_tab[2][(b >> 8) & 0xff] ^ crc_tab[1][(b >> 16) & 0xff] ^ crc_tab[0][b >> 24];
		p += 8; len -= 8;
	}
	while (len--) crc = (crc >> 8) ^ crc_tab[0][(crc ^ *p++) & 0xff];
	return ~crc;
}

/* crc(A || B) from crc(A), crc(B), len(B): multiply crc(A) by x^(8 len(B)) in GF(2)[x] / P */
static uint32_t gf2_times(const uint32_t *mat, uint32_t vec)
{
	uint32_t sum = 0;
	for (; vec; vec >>= 1, mat++)
		if (vec & 1) sum ^= *mat;
	return sum;
}
static void gf2_square(uint32_t *sq, const uint32_t *mat)
{
	for (int n = 0; n < 32; n++) sq[n] = gf2_times(mat, mat[n]);
}
static uint32_t crc_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
	uint32_t even[32], odd[32];
	if (len2 == 0) return crc1;
	odd[0] = 0xEDB88320u;
	uint32_t row = 1;
	for (int n = 1; n < 32; n++) { odd[n] = row; row <<= 1; }
	gf2_square(even, odd);
	gf2_square(odd, even);
	do {
		gf2_square(even, odd);
		if (len2 & 1) crc1 = gf2_times(even, crc1);
		len2 >>= 1;
		if (!len2) break;
		gf2_square(odd, even);
		if (len2 & 1) crc1 = gf2_times(odd, crc1);
		len2 >>= 1;
	} while (len2);
	return crc1 ^ crc2;
}

/* one lazily opened device context for large one-buffer calls, guarded by a mutex (the callers'
 * archives may live on different threads; the reference's only global here is its table, archive_crc32.h:48) */
#define LA_HASH_GPU_MIN   (8u << 20)
#define LA_HASH_GPU_RANGE 65536u
#define LA_HASH_GPU_STAGE (256u << 20)
static pthread_mutex_t dev_mu = PTHREAD_MUTEX_INITIALIZER;
static la_gpu_ctx *dev_ctx;
static int dev_state;	/* 0 untried, 1 open, -1 no device */
static void *dev_buf, *dev_jobs, *dev_out;
static la_hash_job *host_jobs;
static uint32_t *host_out;

static int dev_ready(void)
{
	if (dev_state == 0) {
		const uint32_t nj = LA_HASH_GPU_STAGE / LA_HASH_GPU_RANGE;
		/* the same device as the filters and the ZIP reader (its caller for stored entries) use */
		const char *dv = getenv("LA_GPU_DEVICE");
		const int device = dv != NULL ? atoi(dv) : 0;
		dev_state = -1;
		if (la_gpu_open(device, &dev_ctx) == LA_OK &&
		    la_gpu_malloc(dev_ctx, &dev_buf, LA_HASH_GPU_STAGE) == LA_OK &&
		    la_gpu_malloc(dev_ctx, &dev_jobs, nj * sizeof(la_hash_job)) == LA_OK &&
		    la_gpu_malloc(dev_ctx, &dev_out, nj * sizeof(uint32_t)) == LA_OK &&
		    (host_jobs = (la_hash_job *)malloc(nj * sizeof(la_hash_job))) != NULL &&
		    (host_out = (uint32_t *)malloc(nj * sizeof(uint32_t))) != NULL) {
			dev_state = 1;
		} else if (dev_ctx != NULL) {
			/* a later step failed: nothing of the half-built state stays behind */
			if (dev_buf) (void)la_gpu_free(dev_ctx, dev_buf);
			if (dev_jobs) (void)la_gpu_free(dev_ctx, dev_jobs);
			if (dev_out) (void)la_gpu_free(dev_ctx, dev_out);
			free(host_jobs);
			free(host_out);
			la_gpu_close(dev_ctx);
			dev_ctx = NULL;
			dev_buf = dev_jobs = dev_out = NULL;
			host_jobs = NULL;
			host_out = NULL;
		}
	}
	return dev_state == 1;
}

/* returns 0 and the crc of p[0..len) continued from crc, or -1 when the device path is not available */
static int crc_device(uint32_t *crc, const uint8_t *p, size_t len)
{
	int rc = -1;
	pthread_mutex_lock(&dev_mu);
	if (dev_ready()) {
		uint32_t c = *crc;
		rc = 0;
		while (len && rc == 0) {
			const size_t n = len < LA_HASH_GPU_STAGE ? len : LA_HASH_GPU_STAGE;
			const uint32_t nj = (uint32_t)((n + LA_HASH_GPU_RANGE - 1) / LA_HASH_GPU_RANGE);
			for (uint32_t j = 0; j < nj; j++) {
				host_jobs[j].off = (uint64_t)j * LA_HASH_GPU_RANGE;
				host_jobs[j].len = (uint32_t)(j + 1 < nj ? LA_HASH_GPU_RANGE : n - (size_t)j * LA_HASH_GPU_RANGE);
				host_jobs[j].seed = 0;
			}
			if (la_gpu_memcpy_h2d(dev_ctx, dev_buf, p, n) != LA_OK ||
			    la_gpu_memcpy_h2d(dev_ctx, dev_jobs, host_jobs, nj * sizeof(la_hash_job)) != LA_OK ||
			    la_gpu_crc32_many(dev_ctx, (const uint8_t *)dev_buf, (const la_hash_job *)dev_jobs, nj, (uint32_t *)dev_out) != LA_OK ||
			    la_gpu_memcpy_d2h(dev_ctx, host_out, dev_out, nj * sizeof(uint32_t)) != LA_OK ||
			    la_gpu_sync(dev_ctx) != LA_OK) {
				rc = -1;
				break;
			}
			for (uint32_t j = 0; j < nj; j++)
				c = crc_combine(c, host_out[j], host_jobs[j].len);
			p += n; len -= n;
		}
		if (rc == 0)
			*crc = c;
	}
	pthread_mutex_unlock(&dev_mu);
	return rc;
}

unsigned long la_crc32(unsigned long crc, const void *buf, size_t len)
{
	if (buf == NULL)
		return 0;	/* archive_crc32.h:51-52 */
	uint32_t c = (uint32_t)crc;
	if (len >= LA_HASH_GPU_MIN && crc_device(&c, (const uint8_t *)buf, len) == 0)
		return c;
	return crc_host((uint32_t)crc, (const uint8_t *)buf, len);
}

/* host-only form (tests, and callers that must not touch a device) */
unsigned long la_crc32_host(unsigned long crc, const void *buf, size_t len)
{
	if (buf == NULL)
		return 0;
	return crc_host((uint32_t)crc, (const uint8_t *)buf, len);
}
