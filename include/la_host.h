/*
 * la_host.h -- host-side (plain C) helpers of the GPU read filters: the frame /
 * member walkers that turn a compressed image into the job tables of
 * include/la_gpu.h, and the mapping from device status words back to the
 * reference's return codes and error strings.
 *
 * The walkers restate ONLY the framing of the reference filters (the cheap,
 * sequential pointer chase); every checksum and every decoded byte is produced
 * on the device.
 *   lz4 : libarchive/archive_read_support_filter_lz4.c:289-368 (frame select),
 *         :370-469 (descriptor), :471-613 (block header), :670-721 (legacy)
 *   gzip: libarchive/archive_read_support_filter_gzip.c:128-239 (header),
 *         :398-429 (trailer), :431-511 (member loop)
 */
#ifndef LA_HOST_H
#define LA_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "la_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How the walked region ends (what the reference does AFTER the last indexed unit) */
enum {
	LA_END_EOF = 0,		/* end of input, unrecognised trailing data, or another silent end: rc 0 */
	LA_END_TRUNCATED,	/* "truncated lz4 input" / "truncated gzip input" */
	LA_END_MALFORMED,	/* "malformed lz4 data" (descriptor / block size word) */
	LA_END_MALFORMED_SKIP,	/* "Malformed lz4 data" (skippable frame without a length, lz4.c:349-353) */
	LA_END_EMPTY_FRAME,	/* a frame without blocks: its checksum is verified, then the stream ends (SURVEY F11 i) */
	LA_END_NEED_MORE,	/* window ended inside an item and more input may follow (at_eof == 0) */
	LA_END_GZ_NO_TRAILER,	/* deflate body complete, trailer short: ARCHIVE_FATAL without message (gzip.c:419-421) */
	LA_END_GZ_TOO_LARGE,	/* a gzip member of 4 GiB or more (compressed span or decoded size): beyond the 32-bit member
				 * table of this data plane -- an explicit error, never a wrong byte (the reference streams such
				 * members; a deployment keeps the reference filter for them) */
	LA_END_ZSTD_BAD_MAGIC,	/* bytes that are neither a zstd nor a skippable frame where a frame must start: libzstd's
				 * "Unknown frame descriptor" (zstd.c:226-231) */
	LA_END_ZSTD_BAD_BLOCK	/* the last indexed frame stops at a reserved block type / oversized block: the device names it */
};

typedef struct la_lz4_index {
	la_lz4_block *blocks;
	uint32_t      n_blocks, cap_blocks;
	la_lz4_frame *frames;
	uint32_t      n_frames, cap_frames;
	int           end_kind;		/* LA_END_* */
	uint64_t      consumed;		/* bytes of the image covered by complete items */
	uint64_t      max_out;		/* sum of dst_cap over blocks (upper bound of decoded bytes) */
} la_lz4_index;

/* Walk img[0..len).  at_eof: no more input follows this window.  Returns 0, or
 * -1 on allocation failure.  The index owns its arrays (la_lz4_index_free). */
int  la_lz4_index_build(const uint8_t *img, uint64_t len, int at_eof, la_lz4_index *idx);

/* A frame of independent blocks may be larger than one window: with a resume record the walker
 * indexes the complete blocks it sees, marks the frame LA_LZ4F_OPEN and continues it at the start
 * of the next window (frame record flagged LA_LZ4F_CONT, no descriptor).  Zero the record before
 * the first window.  Legacy frames continue the same way (their blocks share nothing).  In a frame
 * of dependent blocks the first block of a continuation is flagged LA_LZ4B_HIST: the caller must
 * put the tail of the previous window's output in front of the next slab (la_lz4_batch.hist_len). */
typedef struct la_lz4_resume {
	uint32_t in_frame;	/* the next window starts inside a frame (1) / a legacy frame (2) */
	uint32_t bmax;		/* its block maximum */
	uint32_t flags;		/* bit 0: block checksums, bit 1: content checksum, bit 2: dependent blocks */
	uint32_t blocks_so_far;	/* blocks of the frame indexed in earlier windows */
} la_lz4_resume;
int  la_lz4_index_build2(const uint8_t *img, uint64_t len, int at_eof, la_lz4_resume *rs, la_lz4_index *idx);
/* out_budget: stop indexing in front of the block that would take the window's decoded bytes (sum of
 * dst_cap) past it (0 = no bound); the rest of the stream is the next window's */
int  la_lz4_index_build3(const uint8_t *img, uint64_t len, int at_eof, la_lz4_resume *rs, uint64_t out_budget,
         la_lz4_index *idx);
void la_lz4_index_free(la_lz4_index *idx);

/* Bid functions (lz4.c:138-183, gzip.c:244-255) over a peeked buffer */
int  la_lz4_bid_bytes(const uint8_t *p, size_t avail);

/* ---- gzip ---- */
typedef struct la_gz_header {	/* what peek_at_header (gzip.c:128-239) extracts */
	uint64_t off;		/* where the member starts in the image */
	uint32_t len;		/* header bytes */
	uint32_t mtime;
	uint32_t name_off;	/* offset of the FNAME string inside the image, 0 if absent */
	uint32_t bgzf_size;	/* total member size from a BGZF "BC" extra subfield, 0 if absent */
} la_gz_header;

typedef struct la_gz_index {
	la_gz_member *members;	/* src_off/src_len = deflate body + trailer (+ slack when speculative) */
	la_gz_header *headers;
	uint32_t      n, cap;
	int           end_kind;	/* LA_END_* : EOF (silent end / trailing garbage), TRUNCATED, NEED_MORE */
	uint64_t      consumed;	/* bytes of the image covered by the indexed members */
	uint64_t      max_out;	/* sum of dst_cap */
	int           speculative;	/* 1: some boundaries come from the 1f 8b 08 scan and must be confirmed by the decode */
} la_gz_index;

/* Header length (0 = not a gzip header / not enough bytes), gzip.c:128-239 */
size_t la_gz_header_parse(const uint8_t *p, size_t avail, la_gz_header *h);
int    la_gz_bid_bytes(const uint8_t *p, size_t avail);
/* Walk img[0..len): member table with per-member output slots (dst_off assigned back to
 * back from each member's ISIZE claim).  first_only: index just the first member. */
uint64_t la_gz_span_limit(void);	/* 4 GiB - 1 (LA_GZ_TEST_SPAN_LIMIT lowers it for tests) */
int    la_gz_index_build(const uint8_t *img, uint64_t len, int at_eof, la_gz_index *idx);
/* first_skip: candidate boundaries to pass over for the first member (refuted by a decode);
 * first_cap: minimum output slot of the first member (its ISIZE claim proved too small) */
int    la_gz_index_build2(const uint8_t *img, uint64_t len, int at_eof, uint32_t first_skip,
           uint32_t first_cap, la_gz_index *idx);
/* flags: LA_GZ_INDEX_STRICT = a speculative boundary must also carry the XFL / OS bytes real writers emit
 * (XFL 0, 2 or 4; OS 0..13 or 255) -- a few thousand times fewer false boundaries inside deflate data.  A stream
 * whose headers do not look like that is still read correctly: the decode of the member in front ends early, the
 * filter sees a header there and goes on without the flag (la_filter_gzip.c). */
#define LA_GZ_INDEX_STRICT 1u
int    la_gz_index_build3(const uint8_t *img, uint64_t len, int at_eof, uint32_t first_skip,
           uint32_t first_cap, uint32_t flags, la_gz_index *idx);
/* out_budget: stop indexing once the members taken ask for this many decoded bytes (0 = no bound) */
int    la_gz_index_build4(const uint8_t *img, uint64_t len, int at_eof, uint32_t first_skip,
           uint32_t first_cap, uint32_t flags, uint64_t out_budget, la_gz_index *idx);
void   la_gz_index_free(la_gz_index *idx);

/* ---- zstd (host/la_zstd_index.c) ---- */
typedef struct la_zstd_index_result {
	uint32_t n_frames;	/* entries written to frames[] (skippable frames are passed over) */
	int      end_kind;	/* LA_END_EOF, _TRUNCATED, _NEED_MORE, _ZSTD_BAD_MAGIC, _ZSTD_BAD_BLOCK */
	uint64_t consumed;	/* bytes of the image covered by the indexed frames (and the skippable ones between them) */
	uint64_t dst_bytes;	/* decoded bytes the slots ask for (16-byte aligned slots back to back) */
	int      window_full;	/* stopped because of cap / out_budget, not because of the input */
} la_zstd_index_result;
/* zstd.c:107-131 over a peeked buffer */
int la_zstd_bid_bytes(const uint8_t *p, size_t avail);
/* Frame table of img[0..len): frame and block headers only.  out_budget as for the other walkers. */
int la_zstd_index_build(const uint8_t *img, uint64_t len, int at_eof, uint64_t out_budget, la_zstd_frame *frames,
        uint32_t cap, la_zstd_index_result *res);

/* ---- bid policy (host/la_bid_policy.c): does the stream hold many independent units, or ONE serial one that the
 * reference's own filter decodes faster on a host core?  Used by the three bidders; LA_GPU_BID=all switches it off. ---- */
struct archive_read_filter;
int    la_bid_take_all(void);
size_t la_bid_lookahead(size_t default_kib);
const unsigned char *la_bid_peek(struct archive_read_filter *filter, size_t want, size_t *got);
int    la_bid_gzip_parallel(const unsigned char *p, size_t n, size_t hdr_len, size_t lookahead);
int    la_bid_lz4_parallel(const unsigned char *p, size_t n, size_t lookahead);
int    la_bid_zstd_parallel(const unsigned char *p, size_t n, size_t lookahead);

/* ---- hash drop-ins (host/la_hash_dropin.c) ----
 * The 4-pointer table of libarchive/archive_xxhash.h:37-46 (defined as `__archive_xxhash` when built
 * inside libarchive with -DLA_IN_LIBARCHIVE) and crc32() with zlib's signature
 * (libarchive/archive_crc32.h:43-52).  One-buffer XXH32 calls run on the calling thread (a single
 * XXH32 is one serial chain); la_crc32 sends buffers of 8 MiB and more through la_gpu_crc32_many. */
struct la_archive_xxhash {
	unsigned int (*XXH32)(const void *input, unsigned int len, unsigned int seed);
	void        *(*XXH32_init)(unsigned int seed);			/* malloc'ed, free()-able */
	int          (*XXH32_update)(void *state, const void *input, unsigned int len);	/* 0 = XXH_OK */
	unsigned int (*XXH32_digest)(void *state);			/* frees the state */
};
#ifdef LA_IN_LIBARCHIVE
extern const struct la_archive_xxhash __archive_xxhash;
#else
extern const struct la_archive_xxhash la_archive_xxhash;
#endif
unsigned long la_crc32(unsigned long crc, const void *buf, size_t len);
unsigned long la_crc32_host(unsigned long crc, const void *buf, size_t len);

/* The reference's error string for a device status word / an end kind */
const char *la_status_message(uint32_t la_st);
const char *la_end_message(int end_kind, int is_gzip);

#ifdef __cplusplus
}
#endif
#endif
