/*
 * la_oracle.h -- CPU ORACLE for the read-filter decompression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker.  The product path
 * (libarchive_amd/) never links, loads or calls it.
 *
 * It is a from-the-spec restatement, in plain C, of what libarchive's lz4 and
 * gzip read filters compute (reference files cited per function, paths are
 * relative to the reference tree):
 *   - XXH32            libarchive/xxhash.c:234-319 (one shot), :325-507 (streaming)
 *   - CRC32            libarchive/archive_crc32.h:43-84
 *   - LZ4 block decode liblz4 LZ4_decompress_safe[_usingDict] (NOT in the reference
 *                      tree; call sites libarchive/archive_read_support_filter_lz4.c:559,
 *                      :579, :711).  Third-party dependency, the reference pins no
 *                      version (API switch at LZ4_VERSION_MINOR >= 7, lz4.c:578); this
 *                      image carries liblz4 1.9.3, whose accept/reject rules are
 *                      restated here from the published LZ4 block format.
 *   - inflate          zlib inflate() raw deflate, RFC 1951 (NOT in the reference tree;
 *                      call site libarchive/archive_read_support_filter_gzip.c:479;
 *                      reference requires zlib >= 1.2.1, this image carries 1.2.11).
 *   - lz4 / gzip stream framing, return codes and error strings of the two filters:
 *                      archive_read_support_filter_lz4.c:289-721,
 *                      archive_read_support_filter_gzip.c:128-239, :340-511.
 *
 * Parity pinning: the hash functions are checked against the REAL reference code
 * compiled from /root/reference into oracle/_ref (see oracle/Makefile); the codecs
 * and the framing are checked against the reference's own test fixtures
 * (tests/golden/ref_fixtures, digests from SURVEY.md Appendix B), the documented
 * behaviour table and, where the image has them, the system
 * liblz4 / zlib the reference would link.
 */
#ifndef LA_ORACLE_H
#define LA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- XXH32 (xxhash.c:234-319, :325-507) ---- */
uint32_t orc_xxh32(const void *input, size_t len, uint32_t seed);

typedef struct {
	uint64_t total_len;
	uint32_t seed;
	uint32_t v[4];
	uint32_t memsize;
	uint8_t  mem[16];
} orc_xxh32_state;

void     orc_xxh32_init(orc_xxh32_state *st, uint32_t seed);
void     orc_xxh32_update(orc_xxh32_state *st, const void *input, size_t len);
uint32_t orc_xxh32_digest(const orc_xxh32_state *st);

/* ---- CRC32 (archive_crc32.h:43-84); zlib-compatible signature semantics ---- */
uint32_t orc_crc32(uint32_t crc, const void *buf, size_t len);
/* crc of A||B from crc(A), crc(B), len(B) -- GF(2) shift; no reference equivalent,
 * used to check the GPU's wave-parallel reduction. */
uint32_t orc_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

/* ---- LZ4 block (liblz4 LZ4_decompress_safe / _usingDict semantics) ----
 * Returns decoded length (>= 0) or a negative value on malformed input.
 * dict/dict_len: optional prefix that back-references may reach into
 * (dependent blocks, lz4.c:563-584); pass NULL/0 for independent blocks. */
int orc_lz4_block_decode(const uint8_t *src, int src_len,
    uint8_t *dst, int dst_cap, const uint8_t *dict, int dict_len);

/* ---- raw DEFLATE (zlib inflate(), windowBits -15) ----
 * Decodes ONE raw deflate stream starting at src.
 *   *consumed : input bytes used (up to and including the byte holding the last bit)
 *   *produced : output bytes written (also valid on error / truncation: bytes that
 *               zlib would have produced before detecting it)
 * Returns ORC_INF_OK (stream end reached), ORC_INF_TRUNCATED (input ran out
 * mid-stream), ORC_INF_DATA_ERROR (invalid deflate data) or ORC_INF_OUT_FULL. */
enum { ORC_INF_OK = 0, ORC_INF_TRUNCATED = 1, ORC_INF_DATA_ERROR = 2, ORC_INF_OUT_FULL = 3 };
int orc_inflate_raw(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
    size_t *consumed, size_t *produced);

/* ---- stream level: what the reference filter delivers for a whole file ----
 * rc: 0 = ARCHIVE_OK/EOF reached cleanly, -30 = ARCHIVE_FATAL.  errmsg is the
 * exact archive_error_string ("" when the reference sets none).  out receives
 * the concatenation of everything the filter's read() calls returned BEFORE
 * the failing call (the reference drops the partial chunk of a failing call). */
typedef struct {
	int      rc;
	char     errmsg[96];
	size_t   out_len;
	/* gzip only: metadata of the most recently parsed member header at the
	 * moment the first read() returned (SURVEY F11 vi). */
	uint32_t gz_mtime;
	char     gz_name[256];
	int      gz_has_name;
	/* statistics */
	uint64_t n_units;       /* lz4 blocks / gzip members decoded */
	uint64_t n_frames;
	/* new-behaviour verdict, separate from rc (reference never checks, F2) */
	int      gz_trailer_mismatch; /* 1 if some member's CRC32/ISIZE did not match */
} orc_stream_result;

/* bid values (lz4.c:138-183, gzip.c:244-255); 0 = no bid */
int orc_lz4_bid(const uint8_t *p, size_t avail);
/* cpu_baseline only: decode blocks with an external LZ4_decompress_safe / _safe_usingDict (the
 * box's liblz4) behind the same framing and checksums; NULL, NULL restores the port */
void orc_set_external_lz4(void *safe, void *using_dict);
int orc_gzip_bid(const uint8_t *p, size_t avail);

/* Decode a whole .lz4 file image (lz4.c:289-721).  out may be NULL to count only. */
int orc_lz4_stream_decode(const uint8_t *src, size_t src_len,
    uint8_t *out, size_t out_cap, orc_stream_result *res);
/* Decode a whole .gz file image (gzip.c:340-511). */
int orc_gzip_stream_decode(const uint8_t *src, size_t src_len,
    uint8_t *out, size_t out_cap, orc_stream_result *res);

/* gzip header parse (gzip.c:128-239): returns header length or 0. */
size_t orc_gzip_header_len(const uint8_t *p, size_t avail, uint32_t *mtime,
    char *name, size_t name_cap, int *has_name);

#ifdef __cplusplus
}
#endif
/* orc_zstd.c: Zstandard (RFC 8878) restated for the zstd read filter (libarchive/archive_read_support_filter_zstd.c) */
uint64_t orc_xxh64(const void *input, size_t len, uint64_t seed);
int orc_zstd_bid(const uint8_t *p, size_t avail);
int orc_zstd_stream_decode(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *out_len,
    char *msg, size_t msg_cap);

#endif
