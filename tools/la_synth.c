/*
 * la_synth.c -- deterministic synthetic stream generator (bench / test
 * infrastructure; not product, not oracle).
 *
 * Builds the many-block .lz4 stream of SURVEY.md section 8(d): concatenated LZ4
 * frames (magic 0x184D2204, FLG 0x74 = v01 + independent + block checksum +
 * content checksum, BD 0x40 = 64 KiB), `blocks_per_frame` blocks per frame,
 * every block a mix of literal runs (1..24 random bytes) and back-references
 * (length 4..40, offset 1..min(pos,65535)) at match probability 0.55.  The
 * sequences are emitted directly in LZ4 block syntax (that IS the payload
 * model; no compressor is involved), honouring the format's end-of-block
 * rules (last 5 bytes literal, last match starts >= 12 bytes before the end)
 * so that liblz4, the oracle and the GPU decoder all accept every block.
 * The byte layout follows the reference's writer,
 * libarchive/archive_write_add_filter_lz4.c:394-419 (descriptor) and :484-532
 * (block emit), and its reader lz4.c:370-469 / :471-613.
 *
 * Everything is a pure function of (seed, block index), so any block range can
 * be regenerated anywhere (CPU sample checks, multi-GPU shards).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

uint32_t orc_xxh32(const void *input, size_t len, uint32_t seed);	/* oracle/orc_hash.c */

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r)
{
	uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static inline uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)((rng_next(r) >> 11) % n); }

static size_t put_len(uint8_t *c, size_t cp, uint32_t v)
{
	/* v = length - 15, emitted as 255,255,...,rest */
	while (v >= 255) { c[cp++] = 255; v -= 255; }
	c[cp++] = (uint8_t)v;
	return cp;
}

/*
 * One block.  plain (may be NULL... no: required, matches are copied from it)
 * receives `out_size` bytes; comp receives the LZ4 block (worst case
 * out_size + out_size/255 + 16 bytes).  Returns the compressed length.
 */
uint32_t la_synth_lz4_block(uint64_t seed, uint64_t block_index, uint32_t out_size,
    uint8_t *plain, uint8_t *comp)
{
	rng_t r = { seed ^ (block_index * 0xD1342543DE82EF95ull + 0x4C413335ull) };
	uint32_t op = 0;
	size_t cp = 0;
	rng_next(&r);

	for (;;) {
		/* literal items until a match item comes up */
		uint32_t lit = 0;
		while (rng_below(&r, 100) >= 55)
			lit += 1 + rng_below(&r, 24);
		if (op == 0 && lit == 0)
			lit = 1 + rng_below(&r, 24);	/* a block cannot start with a match */
		uint32_t mlen = 4 + rng_below(&r, 37);

		/* does a further (non-final) sequence still fit?  literals must end
		 * <= out_size-12 and the match must end <= out_size-5 */
		if (out_size < 13 || op + lit > out_size - 12 || op + lit + mlen > out_size - 5) {
			uint32_t fin = out_size - op;	/* final literal-only sequence */
			uint8_t tok = (uint8_t)((fin >= 15 ? 15 : fin) << 4);
			comp[cp++] = tok;
			if (fin >= 15) cp = put_len(comp, cp, fin - 15);
			for (uint32_t i = 0; i < fin; i++) {
				uint8_t b = (uint8_t)rng_next(&r);
				plain[op++] = b; comp[cp++] = b;
			}
			break;
		}
		uint32_t ml = mlen - 4;
		comp[cp++] = (uint8_t)(((lit >= 15 ? 15 : lit) << 4) | (ml >= 15 ? 15 : ml));
		if (lit >= 15) cp = put_len(comp, cp, lit - 15);
		for (uint32_t i = 0; i < lit; i++) {
			uint8_t b = (uint8_t)rng_next(&r);
			plain[op++] = b; comp[cp++] = b;
		}
		uint32_t maxoff = op < 65535 ? op : 65535;
		uint32_t off = 1 + rng_below(&r, maxoff);
		comp[cp++] = (uint8_t)off; comp[cp++] = (uint8_t)(off >> 8);
		if (ml >= 15) cp = put_len(comp, cp, ml - 15);
		for (uint32_t i = 0; i < mlen; i++, op++)
			plain[op] = plain[op - off];
	}
	return (uint32_t)cp;
}

static void put32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* upper bound of one frame's size */
uint64_t la_synth_lz4_frame_bound(uint32_t blocks_per_frame, uint32_t block_size)
{
	return 7 + (uint64_t)blocks_per_frame * (4 + block_size + block_size / 255 + 16 + 4) + 8;
}

/*
 * One frame = blocks [first_block, first_block + nblocks).  Returns frame size.
 * plain (nblocks*block_size bytes) receives the decoded payload.
 */
uint64_t la_synth_lz4_frame(uint64_t seed, uint64_t first_block, uint32_t nblocks,
    uint32_t block_size, uint8_t *plain, uint8_t *out)
{
	uint64_t o = 0;
	put32(out + o, 0x184D2204u); o += 4;
	out[o] = 0x74;		/* v01 | independent | block checksum | content checksum */
	out[o + 1] = 0x40;	/* 64 KiB; (block_size is the decoded size used, <= 64 KiB) */
	out[o + 2] = (uint8_t)((orc_xxh32(out + o, 2, 0) >> 8) & 0xff);
	o += 3;
	for (uint32_t b = 0; b < nblocks; b++) {
		uint32_t cl = la_synth_lz4_block(seed, first_block + b, block_size,
		    plain + (uint64_t)b * block_size, out + o + 4);
		put32(out + o, cl);
		put32(out + o + 4 + cl, orc_xxh32(out + o + 4, cl, 0));
		o += 4 + cl + 4;
	}
	put32(out + o, 0); o += 4;
	put32(out + o, orc_xxh32(plain, (uint64_t)nblocks * block_size, 0)); o += 4;
	return o;
}

typedef struct {
	uint64_t seed, first_frame, nframes;
	uint32_t bpf, bs;
	uint8_t *plain;		/* nframes*bpf*bs, or NULL (then a per-thread scratch) */
	uint8_t *stage;		/* nframes * bound */
	uint64_t *sizes;	/* nframes */
	uint64_t bound;
} job_t;

static void *worker(void *arg)
{
	job_t *j = (job_t *)arg;
	uint8_t *scratch = NULL;
	if (!j->plain)
		scratch = malloc((size_t)j->bpf * j->bs);
	for (uint64_t f = 0; f < j->nframes; f++) {
		uint8_t *pl = j->plain ? j->plain + f * j->bpf * j->bs : scratch;
		j->sizes[f] = la_synth_lz4_frame(j->seed, (j->first_frame + f) * j->bpf, j->bpf, j->bs,
		    pl, j->stage + f * j->bound);
	}
	free(scratch);
	return NULL;
}

/*
 * Build `nframes` frames starting at frame index `first_frame` into `out`
 * (capacity out_cap), using `nthreads` threads.  plain may be NULL.  Returns
 * the stream length, or 0 if out_cap is too small / allocation failed.
 */
uint64_t la_synth_lz4_stream(uint64_t seed, uint64_t first_frame, uint64_t nframes,
    uint32_t blocks_per_frame, uint32_t block_size, int nthreads,
    uint8_t *plain, uint8_t *out, uint64_t out_cap)
{
	uint64_t bound = la_synth_lz4_frame_bound(blocks_per_frame, block_size);
	if (nthreads < 1) nthreads = 1;
	if ((uint64_t)nthreads > nframes) nthreads = (int)(nframes ? nframes : 1);
	uint8_t *stage = malloc(nframes * bound);
	uint64_t *sizes = malloc(nframes * sizeof(uint64_t));
	pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nthreads);
	job_t *jobs = malloc(sizeof(job_t) * (size_t)nthreads);
	if (!stage || !sizes || !th || !jobs) { free(stage); free(sizes); free(th); free(jobs); return 0; }
	uint64_t per = (nframes + (uint64_t)nthreads - 1) / (uint64_t)nthreads, done = 0;
	int nt = 0;
	for (int t = 0; t < nthreads && done < nframes; t++, nt++) {
		uint64_t n = nframes - done < per ? nframes - done : per;
		jobs[t] = (job_t){ seed, first_frame + done, n, blocks_per_frame, block_size,
		    plain ? plain + done * blocks_per_frame * block_size : NULL,
		    stage + done * bound, sizes + done, bound };
		pthread_create(&th[t], NULL, worker, &jobs[t]);
		done += n;
	}
	for (int t = 0; t < nt; t++)
		pthread_join(th[t], NULL);
	uint64_t o = 0;
	for (uint64_t f = 0; f < nframes; f++) {
		if (o + sizes[f] > out_cap) { o = 0; break; }
		memcpy(out + o, stage + f * bound, sizes[f]);
		o += sizes[f];
	}
	free(stage); free(sizes); free(th); free(jobs);
	return o;
}
