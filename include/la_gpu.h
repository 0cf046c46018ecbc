/*
 * la_gpu.h -- C ABI of the MI355X (gfx950) data plane behind libarchive's
 * lz4 / gzip read filters.
 *
 * This is the `extern "C"` shim the host filters (include/la_filter.h,
 * libarchive_amd/host/) call from inside their vtable read() -- the ONLY place
 * a device boundary is crossed.  Plain pointers and sizes,
 * no C++ or framework types.  Each entry point cites the reference code whose
 * arithmetic it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - Functions return LA_OK (0) or a negative la_rc.  Nothing throws.
 *   - Pointers named d_* are DEVICE pointers; h_* are host pointers.
 *   - Work is enqueued on the context's HIP stream; results are valid after
 *     la_gpu_sync() (or after the caller synchronises that stream itself).
 *   - A context is owned by one host thread (one `struct archive` = one
 *     thread, reference README.md:194-221); any number of contexts may exist.
 */
#ifndef LA_GPU_H
#define LA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LA_GPU_ABI_VERSION 3

typedef enum la_rc {
	LA_OK            = 0,
	LA_ERR_NO_DEVICE = -1,	/* no usable gfx950 device / HIP runtime failure at open */
	LA_ERR_HIP       = -2,	/* a HIP call failed; see la_gpu_last_error() */
	LA_ERR_ARG       = -3,
	LA_ERR_NOMEM     = -4
} la_rc;

typedef struct la_gpu_ctx la_gpu_ctx;

/* ---- context ---- */
int         la_gpu_abi_version(void);
int         la_gpu_device_count(void);
/* Opens device `device`, creates a private stream and a small workspace. */
int         la_gpu_open(int device, la_gpu_ctx **out);
void        la_gpu_close(la_gpu_ctx *ctx);
/* Run on a caller-owned hipStream_t instead of the private one (NULL = back to private). */
int         la_gpu_set_stream(la_gpu_ctx *ctx, void *hip_stream);
int         la_gpu_sync(la_gpu_ctx *ctx);
const char *la_gpu_last_error(const la_gpu_ctx *ctx);
/* Pre-size the context's device workspace (sequence tables, scan scratch) so that
 * no allocation happens inside a decode call.  Optional. */
int         la_gpu_reserve(la_gpu_ctx *ctx, uint64_t workspace_bytes);

/* Raw device memory helpers for C hosts that do not own an allocator. */
int         la_gpu_malloc(la_gpu_ctx *ctx, void **d_ptr, uint64_t bytes);
int         la_gpu_free(la_gpu_ctx *ctx, void *d_ptr);
int         la_gpu_malloc_host(la_gpu_ctx *ctx, void **h_ptr, uint64_t bytes);	/* pinned */
int         la_gpu_free_host(la_gpu_ctx *ctx, void *h_ptr);
int         la_gpu_memcpy_h2d(la_gpu_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int         la_gpu_memcpy_d2h(la_gpu_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int         la_gpu_memcpy_d2d(la_gpu_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes);

/* A marker in the stream: la_gpu_mark() notes the point reached so far, la_gpu_wait_mark()
 * blocks the host until everything queued BEFORE the marker is done -- work queued after it
 * keeps running (how the filters wait for a slab copy while the next window decodes). */
int         la_gpu_mark(la_gpu_ctx *ctx);
int         la_gpu_wait_mark(la_gpu_ctx *ctx);

/* Stream-ordered timer (HIP events on the context's stream). */
int         la_gpu_timer_start(la_gpu_ctx *ctx);
int         la_gpu_timer_stop(la_gpu_ctx *ctx, float *elapsed_ms);	/* synchronises */

/* Per-phase timing of the most recent batch call (HIP events between the kernels,
 * on the work stream).  la_gpu_profile_read() synchronises on the last event and
 * returns the number of phases filled in (0 when profiling is off). */
#define LA_PROF_MAX_PHASES 12
int         la_gpu_profile_enable(la_gpu_ctx *ctx, int on);
int         la_gpu_profile_read(la_gpu_ctx *ctx, float *ms, const char **names, int cap);

/* ---- per-unit status words written by the device ---- */
enum {
	LA_ST_OK                  = 0,
	LA_ST_LZ4_BAD_BLOCK_SUM   = 1,	/* lz4.c:517-526  -> "malformed lz4 data" */
	LA_ST_LZ4_DECODE          = 2,	/* lz4.c:594-598  -> "lz4 decompression failed" */
	LA_ST_LZ4_BAD_HEADER_SUM  = 3,	/* lz4.c:446-451  -> "malformed lz4 data" */
	LA_ST_LZ4_BAD_CONTENT_SUM = 4,	/* lz4.c:655-660  -> "lz4 stream checksum error" */
	LA_ST_GZ_DATA             = 5,	/* gzip.c:494-499 -> "gzip decompression failed" */
	LA_ST_GZ_TRUNCATED        = 6,	/* gzip.c:464-469 -> "truncated gzip input" */
	LA_ST_GZ_BAD_CRC          = 7,	/* NEW (reference never checks, gzip.c:423) */
	LA_ST_GZ_BAD_ISIZE        = 8,	/* NEW */
	LA_ST_GZ_OUT_FULL         = 9,	/* member produced more than dst_cap bytes: host retries with a larger slot */
	LA_ST_GZ_NO_TRAILER       = 10,	/* deflate body complete, fewer than 8 trailer bytes inside src_len (gzip.c:419-421) */
	/* zstd.c:226-231 -> "Zstd decompression failed: <libzstd's name of the error>" */
	LA_ST_ZSTD_CORRUPT        = 11,	/* "Corrupted block detected" */
	LA_ST_ZSTD_TRUNCATED      = 12,	/* the frame needs more bytes than src_len (zstd.c:213-217 "Truncated zstd input") */
	LA_ST_ZSTD_BAD_CHECKSUM   = 13,	/* "Restored data doesn't match checksum" */
	LA_ST_ZSTD_OUT_FULL       = 14,	/* the frame produces more than dst_cap bytes: host retries with a larger slot */
	LA_ST_ZSTD_UNSUPPORTED    = 15,	/* reserved header bit: "Unsupported frame parameter" */
	LA_ST_ZSTD_WINDOW         = 16,	/* window above 2^27 (ZSTD_decompressStream's default limit): "Frame requires too much memory for decoding" */
	LA_ST_ZSTD_DICTIONARY     = 17	/* the frame names a dictionary: "Dictionary mismatch" */
};

/* =====================================================================
 * XXH32 -- replaces __archive_xxhash.XXH32 (libarchive/xxhash.c:234-319) for
 * MANY independent hashes at once (one hash is a serial chain).
 * ===================================================================== */
typedef struct la_hash_job {
	uint64_t off;	/* byte offset of the range inside d_base */
	uint32_t len;
	uint32_t seed;
} la_hash_job;

int la_gpu_xxh32_many(la_gpu_ctx *ctx, const uint8_t *d_base,
    const la_hash_job *d_jobs, uint32_t n_jobs, uint32_t *d_out);

/* =====================================================================
 * CRC32 -- replaces crc32() (libarchive/archive_crc32.h:43-84, zlib-compatible)
 * for many ranges; each range is reduced wave-parallel with GF(2) combines.
 * `seed` of a job is the running crc to continue from (0 for a fresh one).
 * ===================================================================== */
int la_gpu_crc32_many(la_gpu_ctx *ctx, const uint8_t *d_base,
    const la_hash_job *d_jobs, uint32_t n_jobs, uint32_t *d_out);

/* =====================================================================
 * LZ4 -- replaces the data plane of lz4_filter_read_data_block /
 * lz4_filter_read_default_stream / _legacy_stream
 * (libarchive/archive_read_support_filter_lz4.c:471-613, :615-668, :670-721):
 * block checksum XXH32, LZ4_decompress_safe[_usingDict], content checksum,
 * header check byte -- for a whole batch of blocks/frames per call.
 * The host walks the size words (cheap pointer chase) and fills these tables.
 * ===================================================================== */
#define LA_LZ4B_STORED    1u	/* size word had bit 31: payload is the data (lz4.c:500-504, :530-552) */
#define LA_LZ4B_CHECKSUM  2u	/* block_sum holds the LE32 that followed the payload (lz4.c:517-526) */
#define LA_LZ4B_DEPENDENT 4u	/* frame without the independence bit: matches may reach the previous block (lz4.c:562-591) */
#define LA_LZ4B_FIRST     8u	/* first block of its frame: dictionary is 64 KiB of zeros (lz4.c:260-261) */
#define LA_LZ4B_HIST     16u	/* dependent block that continues a frame from the previous batch: its dictionary is
					 * the la_lz4_batch.hist_len bytes in FRONT of d_dst (d_dst[-hist_len .. 0)) */

typedef struct la_lz4_block {
	uint64_t src_off;	/* first payload byte inside d_src (after the 4-byte size word) */
	uint32_t src_len;	/* payload bytes (size word & 0x7fffffff) */
	uint32_t dst_cap;	/* frame's block maximum size, or 8 MiB for legacy blocks */
	uint32_t flags;		/* LA_LZ4B_* */
	uint32_t block_sum;	/* expected XXH32 of the payload when LA_LZ4B_CHECKSUM */
} la_lz4_block;

#define LA_LZ4F_CONTENT_SUM 1u	/* content_sum holds the LE32 after the EndMark (lz4.c:639-662) */
#define LA_LZ4F_HEADER_SUM  2u	/* verify descriptor check byte (lz4.c:446-451) */
#define LA_LZ4F_CONT        4u	/* the frame began in an earlier batch: no descriptor here, its content hash
					 * continues from d_carry_in */
#define LA_LZ4F_OPEN        8u	/* the frame goes on in the next batch: its content hash state goes to d_carry_out */
#define LA_LZ4F_HASHED     16u	/* the frame carries a content checksum (set on every piece of such a frame) */

typedef struct la_lz4_frame {
	uint64_t desc_off;	/* offset of FLG inside d_src */
	uint32_t desc_len;	/* descriptor bytes INCLUDING the trailing check byte (3..15) */
	uint32_t first_block;	/* index of the frame's first block in the block table */
	uint32_t n_blocks;
	uint32_t flags;		/* LA_LZ4F_* */
	uint32_t content_sum;	/* expected XXH32 of the frame's decoded bytes */
	uint32_t reserved;
} la_lz4_frame;

/* Batch summary reduced on the device (first failing event in STREAM order). */
typedef struct la_batch_summary {
	uint64_t total_out;		/* decoded bytes of the whole batch */
	uint32_t n_bad_units;		/* blocks / members with status != 0 */
	uint32_t n_bad_frames;
	uint32_t first_bad_unit;	/* 0xFFFFFFFF if none */
	uint32_t first_bad_frame;	/* 0xFFFFFFFF if none */
	uint32_t first_zero_unit;	/* first unit that decoded to 0 bytes (ends the stream, SURVEY F11 i); 0xFFFFFFFF if none */
	uint32_t reserved;
} la_batch_summary;

#define LA_LZ4_OPT_GENERAL_ONLY 1u	/* force the general (any block size / dependent) expand kernel */
#define LA_LZ4_OPT_NO_VERIFY    2u	/* skip the three XXH32 checks (the reference's `!stream-checksum` shape) */
#define LA_LZ4_OPT_PARSE_V1     4u	/* first-generation parse: block checksums and token walk as two kernels
					 * reading global memory per lane (kept as a cross-check of the staged one) */

#define LA_LZ4_OPT_EXPAND_INORDER 8u	/* second implementation of the LDS-window expand step (la_lz4_inorder.hip, round 3: one
					 * matcher wave takes the sequences in stream order, literal waves run ahead of it, a flush wave
					 * behind it; no per-sequence flags, no polling in the match phase); same results, kept as the
					 * cross-check of the default kernel (la_lz4_fast.hip), which is still the faster one */

typedef struct la_lz4_batch {
	const uint8_t      *d_src;	/* compressed image (or batch window) in HBM */
	uint64_t            src_bytes;
	const la_lz4_block *d_blocks;
	uint32_t            n_blocks;
	const la_lz4_frame *d_frames;	/* may be NULL when n_frames == 0 */
	uint32_t            n_frames;
	uint8_t            *d_dst;	/* decoded slab: blocks are packed back to back in table order */
	uint64_t            dst_cap;
	/* outputs */
	uint32_t           *d_out_len;		/* [n_blocks]  decoded bytes per block */
	uint64_t           *d_dst_off;		/* [n_blocks+1] exclusive prefix sums of d_out_len */
	uint32_t           *d_block_status;	/* [n_blocks]  LA_ST_* */
	uint32_t           *d_frame_status;	/* [n_frames]  LA_ST_* */
	la_batch_summary   *d_summary;		/* one record */
	uint32_t            options;		/* LA_LZ4_OPT_* */
	uint32_t            hist_len;		/* bytes of carried-over output in front of d_dst (LA_LZ4B_HIST), else 0 */
	/* XXH32 state of a content checksum that spans batches (one frame at most enters a batch
	 * and one at most leaves it unfinished): LA_XXH_CARRY_BYTES each, may be NULL when no
	 * frame has LA_LZ4F_CONT / LA_LZ4F_OPEN.  Must be two different buffers. */
	const void         *d_carry_in;
	void               *d_carry_out;
} la_lz4_batch;
#define LA_XXH_CARRY_BYTES 64u

/* Workspace bytes la_gpu_lz4_decode() needs for this shape (for la_gpu_reserve). */
uint64_t la_gpu_lz4_workspace_bytes(uint32_t n_blocks, uint64_t src_bytes);

int la_gpu_lz4_decode(la_gpu_ctx *ctx, const la_lz4_batch *batch);

/* =====================================================================
 * gzip / DEFLATE -- replaces the inflate() loop of gzip_filter_read
 * (libarchive/archive_read_support_filter_gzip.c:431-511; zlib inflate with
 * windowBits -15) for a batch of independent members, plus the trailer
 * CRC32/ISIZE check the reference leaves as a TODO (gzip.c:423).
 * ===================================================================== */
typedef struct la_gz_member {
	uint64_t src_off;	/* first byte of the raw deflate body inside d_src */
	uint32_t src_len;	/* bytes available for body + trailer (up to the next member / end) */
	uint32_t dst_cap;	/* capacity reserved for this member's output */
	uint64_t dst_off;	/* where its output goes inside d_dst */
} la_gz_member;

typedef struct la_gz_result {
	uint32_t status;	/* LA_ST_* (a CRC/ISIZE mismatch is reported here but the bytes are still delivered) */
	uint32_t out_len;	/* bytes produced (also on error: what zlib would have emitted) */
	uint32_t consumed;	/* deflate body bytes consumed (trailer follows) */
	uint32_t crc32;		/* CRC32 of the produced bytes */
} la_gz_result;

typedef struct la_gz_batch {
	const uint8_t      *d_src;
	uint64_t            src_bytes;
	const la_gz_member *d_members;
	uint32_t            n_members;
	uint8_t            *d_dst;
	uint64_t            dst_cap;
	la_gz_result       *d_results;	/* [n_members] */
	la_batch_summary   *d_summary;
	uint32_t            options;
	uint32_t            reserved;
} la_gz_batch;

#define LA_GZ_OPT_NO_VERIFY   1u	/* do not compare the trailer (reference behaviour) */
#define LA_GZ_OPT_WAVE_KERNEL 2u	/* force the wave-per-member kernel (default: lane-per-member from 8192 members up) */
#define LA_GZ_OPT_LANE_KERNEL 4u	/* force the in-place lane-per-member kernel */
#define LA_GZ_OPT_TWO_PHASE   8u	/* force entropy decode + LDS-window expand (the default from 8192 members up) */

#define LA_GZ_OPT_EXPAND_INORDER 32u	/* two-phase path: build the output with the in-order expand kernel (cross-check) */
#define LA_GZ_OPT_RAW        16u	/* members are bare raw-deflate streams (ZIP entries, archive_read_support_format_zip.c:2536-2700):
					 * no gzip trailer follows the body, nothing is compared; status, out_len, consumed and
					 * the CRC32 of the produced bytes are reported */

int la_gpu_gzip_decode(la_gpu_ctx *ctx, const la_gz_batch *batch);

/* =====================================================================
 * Zstandard -- replaces, for a batch of whole frames per call, the ZSTD_decompressStream loop of
 * zstd_filter_read (libarchive/archive_read_support_filter_zstd.c:171-260; libzstd is the reference's
 * external dependency for this codec): frame header, raw / RLE / compressed blocks, XXH64 content checksum.
 * The host walks the frame and block headers (la_zstd_index_build, include/la_host.h), drops skippable
 * frames and fills this table; one frame = one unit (a frame is one serial chain).
 * ===================================================================== */
typedef struct la_zstd_frame {
	uint64_t src_off;	/* the frame's magic number inside d_src */
	uint64_t src_len;	/* bytes of the whole frame (header .. last block / checksum) */
	uint64_t dst_off;	/* where its output goes inside d_dst */
	uint64_t dst_cap;	/* capacity reserved for it: Frame_Content_Size when the header carries one, else the walker's bound */
} la_zstd_frame;

typedef struct la_zstd_result {
	uint32_t status;	/* LA_ST_* */
	uint32_t reserved;
	uint64_t out_len;	/* bytes produced (0 on error) */
} la_zstd_result;

typedef struct la_zstd_batch {
	const uint8_t       *d_src;
	uint64_t             src_bytes;
	const la_zstd_frame *d_frames;
	uint32_t             n_frames;
	uint32_t             options;	/* LA_ZSTD_OPT_* */
	uint8_t             *d_dst;
	uint64_t             dst_cap;
	la_zstd_result      *d_results;	/* [n_frames] */
} la_zstd_batch;

#define LA_ZSTD_OPT_NO_VERIFY 1u	/* skip the content checksum */
#define LA_ZSTD_OPT_LANE_KERNEL 2u	/* first-generation kernel, one LANE per frame (default: one wave per frame); same results, kept as a cross-check */

uint64_t la_gpu_zstd_workspace_bytes(uint32_t n_frames);
int      la_gpu_zstd_decode(la_gpu_ctx *ctx, const la_zstd_batch *batch);

/* =====================================================================
 * LZ4 compression -- the data plane of the lz4 WRITE filter: replaces, for a whole
 * stream per call, LZ4_compress_default per independent block, the stored-block fallback, the block
 * checksum, the frame descriptor with its check byte, the EndMark and the content checksum of
 * libarchive/archive_write_add_filter_lz4.c:394-447, :484-532.  d_src[0, src_bytes) is cut into blocks
 * of block_size bytes (at most 64 KiB), blocks_per_frame of them form one frame; d_out receives the
 * concatenated frames, *d_out_bytes their total size (if it exceeds out_cap nothing past out_cap was
 * written: call again with a larger buffer; src_bytes + src_bytes/255 + 27 bytes per block always fit).
 * The bytes are not liblz4's (an LZ4 stream is not unique); every conforming decoder returns the input.
 * ===================================================================== */
#define LA_LZ4C_BLOCK_SUM   1u	/* FLG bit 4: XXH32 of every block's payload as written */
#define LA_LZ4C_CONTENT_SUM 2u	/* FLG bit 2: XXH32 of every frame's input bytes after its EndMark */

typedef struct la_lz4c_batch {
	const uint8_t *d_src;
	uint64_t       src_bytes;
	uint32_t       block_size;		/* 1 .. 65536 */
	uint32_t       blocks_per_frame;	/* >= 1 */
	uint32_t       flags;			/* LA_LZ4C_* */
	uint32_t       reserved;
	uint8_t       *d_out;
	uint64_t       out_cap;
	uint64_t      *d_out_bytes;		/* one u64 on the device */
} la_lz4c_batch;

uint64_t la_gpu_lz4_compress_workspace_bytes(uint64_t src_bytes, uint32_t block_size, uint32_t blocks_per_frame);
/* upper bound of the stream la_gpu_lz4_compress writes for this shape */
uint64_t la_gpu_lz4_compress_bound(uint64_t src_bytes, uint32_t block_size, uint32_t blocks_per_frame);
int      la_gpu_lz4_compress(la_gpu_ctx *ctx, const la_lz4c_batch *batch);

/* =====================================================================
 * gzip compression -- the data plane of the gzip WRITE filter: replaces, for a whole stream per
 * call, deflate() through zlib, the hand-built header, the CRC32 of the input and the trailer of
 * libarchive/archive_write_add_filter_gzip.c:201-237, :263-266, :293-345.  d_src[0, src_bytes) is cut into chunks of
 * chunk_bytes (at most 49152); every chunk becomes one gzip member (fixed-Huffman deflate, or a stored block when
 * it would not shrink) whose header carries the BGZF-compatible "BC" size subfield.  d_out receives the concatenated
 * members, *d_out_bytes their size (beyond out_cap nothing is written; la_gpu_gzip_compress_bound() always fits).
 * ===================================================================== */
typedef struct la_gzc_batch {
	const uint8_t *d_src;
	uint64_t       src_bytes;
	uint32_t       chunk_bytes;	/* 1 .. 49152 */
	uint32_t       mtime;		/* MTIME of every member header (0 = none, gzip.c:213-220 writes time(NULL) unless "!timestamp") */
	uint8_t       *d_out;
	uint64_t       out_cap;
	uint64_t      *d_out_bytes;	/* one u64 on the device */
} la_gzc_batch;

uint64_t la_gpu_gzip_compress_workspace_bytes(uint64_t src_bytes, uint32_t chunk_bytes);
uint64_t la_gpu_gzip_compress_bound(uint64_t src_bytes, uint32_t chunk_bytes);
int      la_gpu_gzip_compress(la_gpu_ctx *ctx, const la_gzc_batch *batch);

#ifdef __cplusplus
}
#endif
#endif /* LA_GPU_H */
