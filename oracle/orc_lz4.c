/*
 * orc_lz4.c -- ORACLE (test infrastructure, see la_oracle.h): LZ4 block decode.
 *
 * The arithmetic is NOT in the reference tree: libarchive calls liblz4's
 * LZ4_decompress_safe (archive_read_support_filter_lz4.c:559, :711) and
 * LZ4_decompress_safe_usingDict (:579).  This restates the published LZ4 block
 * format (sequence = token, literal-length extension, literals, LE16 offset,
 * match-length extension; minmatch 4) with the accept/reject rules of
 * liblz4 1.9.3's safe decoder, the version this image would link:
 *   - the input must be consumed exactly; empty input is an error;
 *   - a sequence whose literals end within 12 bytes of the END OF THE
 *     DESTINATION CAPACITY, or within 8 bytes of the end of input, must be the
 *     last one (its literals must end exactly at the end of input);
 *   - a match may not end within 5 bytes of the end of the capacity;
 *   - an offset that reaches before the start of (dictionary + output) is an error;
 *   - length-extension bytes are bounded as in the library (literal run: the
 *     first extension byte must lie before iend-15; match: every extension byte
 *     must leave ip < iend-4).
 * One deliberate difference: offset 0 is rejected (the format forbids it;
 * liblz4 1.9.3 accepts it and emits indeterminate bytes -- SURVEY Appendix D,
 * "parity unpinned").
 */
#include "la_oracle.h"
#include <string.h>

#define MINMATCH      4
#define MFLIMIT       12
#define LASTLITERALS  5

int orc_lz4_block_decode(const uint8_t *src, int src_len,
    uint8_t *dst, int dst_cap, const uint8_t *dict, int dict_len)
{
	long ip = 0, op = 0;
	const long iend = src_len, oend = dst_cap;

	if (src == NULL || src_len <= 0 || dst_cap < 0)
		return -1;
	if (dst_cap == 0)
		return (src_len == 1 && src[0] == 0) ? 0 : -1;
	if (dict == NULL)
		dict_len = 0;

	for (;;) {
		unsigned token = src[ip++];
		long length = token >> 4;

		if (length == 15) {
			unsigned s;
			if (ip >= iend - 15)
				return -1;
			do {
				s = src[ip++];
				length += s;
				if (ip >= iend - 15)
					break;	/* the library's loop check only stops the loop */
			} while (s == 255);
		}

		if (op + length > oend - MFLIMIT || ip + length > iend - (2 + 1 + LASTLITERALS)) {
			/* must be the last sequence */
			if (ip + length != iend || op + length > oend)
				return -1;
			memmove(dst + op, src + ip, (size_t)length);
			op += length;
			break;
		}
		memcpy(dst + op, src + ip, (size_t)length);
		ip += length;
		op += length;

		long offset = (long)src[ip] | ((long)src[ip + 1] << 8);
		ip += 2;
		length = token & 15;
		if (length == 15) {
			unsigned s;
			do {
				s = src[ip++];
				length += s;
				if (ip >= iend - LASTLITERALS + 1)
					return -1;
			} while (s == 255);
		}
		length += MINMATCH;

		if (offset == 0)
			return -1;	/* see header: deliberate */
		if (offset > op + dict_len)
			return -1;
		if (op + length > oend - LASTLITERALS)
			return -1;

		if (offset >= length && offset <= op) {
			/* no overlap, source entirely inside this block's output */
			memcpy(dst + op, dst + op - offset, (size_t)length);
			op += length;
			continue;
		}
		/* byte-wise copy: overlapping matches replicate */
		for (long i = 0; i < length; i++) {
			long from = op - offset;
			uint8_t b = (from >= 0) ? dst[from] : dict[dict_len + from];
			dst[op++] = b;
		}
	}
	return (int)op;
}
