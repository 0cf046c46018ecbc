/*
 * la_format_zip.c -- ZIP reader with the per-entry deflate + CRC32 on the device
 * (SURVEY 8f-1): the place libarchive really runs raw inflate per entry with an ENFORCED
 * CRC32 behind archive_read_next_header / archive_read_data_block.
 *
 * What it follows (libarchive/archive_read_support_format_zip.c, file:line of the reference):
 *   bid                          :3346-3379  ("PK" + one of six 16-bit tags = 29)
 *   central directory            :3900-4100  (end record, Zip64 locator / record, 46-byte file headers, Zip64
 *                                            extra field 0x0001, correction for data in front of the archive)
 *   local header -> data offset  :930-1020   (30 bytes + name + extra)
 *   entry metadata               :961 (DOS time -> mktime), :1060-1130 (mode from the external attributes of
 *                                            Unix-made entries, else 0664 / 0775 with a trailing '/' = directory)
 *   stored data                  :2810-2880  (zip_read_data_none)
 *   deflate data                 :2536-2700  (zip_read_data_deflate: inflateInit2(-15), inflate, end of entry)
 *   check values                 :3155-3196  (CRC32 "ZIP bad CRC: 0x%lx should be 0x%lx", compressed and
 *                                            uncompressed size checks; ARCHIVE_FAILED)
 *   unsupported methods          :1180-1215, :3125-3147 ("Unsupported ZIP compression method (%d: %s)")
 *
 * Shape.  The reference pulls one entry at a time through zlib.  Here the central directory is walked
 * on the host (entries are independently addressable through it), the archive image is uploaded once,
 * and entries are decoded in BATCHES: every deflate entry of a batch becomes one raw-deflate member of
 * la_gpu_gzip_decode (include/la_gpu.h, LA_GZ_OPT_RAW: no gzip trailer behind the body), which also
 * returns the CRC32 of what it produced; the host compares it with the directory's value.  Stored entries
 * are handed out zero-copy from the gathered image, their CRC32 through la_crc32 (host/la_hash_dropin.c).
 * read_data returns a whole entry per call (any size is legal).
 *
 * The read core of this slice has no seek, so the archive is gathered into memory first (bounded by
 * LA_ZIP_MAX_MIB, default 16 GiB) and the format is recognised by its first bytes only: self-extracting
 * or padded archives, which the reference finds by seeking to the end record, are not opened here.
 * Out of this slice: encryption, methods other than 0 and 8 (reported per entry like the reference does),
 * character-set conversion of names, symlink targets, mac metadata.
 */
#include <errno.h>
#include <inttypes.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "la_read_private.h"
#include "../../include/la_gpu.h"
#include "../../include/la_host.h"

#define ZIP_ENCRYPTED        (1 << 0)
#define ZIP_STRONG_ENCRYPTED (1 << 6)
#define ZIP_BATCH_OUT        (1ull << 30)	/* decoded bytes per device batch */
#define ZIP_BATCH_ENTRIES    262144u

struct zip_entry {
	uint64_t lho, data_off, csize, usize;
	uint64_t slab_off;		/* where its decoded bytes sit in the batch slab */
	uint32_t crc;
	uint64_t name_off;	/* offset of the name inside the gathered image (the central directory may lie beyond 4 GiB) */
	int64_t mtime;
	uint16_t method, flags, name_len;
	unsigned mode;
	uint8_t system;			/* "version made by" high byte: 3 = Unix */
	uint8_t version;		/* version needed to extract */
};

/* (offsets into the gathered image are 64-bit: a Zip64 central directory lies beyond 4 GiB) */
_Static_assert(sizeof(((struct zip_entry *)0)->name_off) == 8, "zip_entry.name_off must hold an offset into an image of up to LA_ZIP_MAX_MIB");

struct zip_private {
	la_gpu_ctx *gpu;
	uint8_t *img;			/* the whole archive */
	size_t img_len, img_cap;
	void *d_img;			/* ... on the device */
	struct zip_entry *ents;
	uint32_t n, cur;		/* cur: entry the last read_header returned (n = none yet) */
	int loaded, started, data_done;
	/* current batch: entries [b_lo, b_hi) decoded into slab */
	uint32_t b_lo, b_hi;
	uint8_t *slab;			/* pinned */
	size_t slab_cap;
	la_gz_result *res;		/* [b_hi - b_lo], index by position among the batch's deflate entries */
	uint32_t *res_idx;		/* [n] entry -> index into res, 0xFFFFFFFF when not a deflate entry of the batch */
	char fmt_name[64];
};

static uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t le32(const uint8_t *p) { return archive_le32dec(p); }
static uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

static int zip_bid(struct archive_read *a, int best_bid)
{
	(void)best_bid;
	const char *p = __archive_read_ahead(a, 4, NULL);
	if (p == NULL)
		return -1;
	if (p[0] == 'P' && p[1] == 'K') {
		if ((p[2] == '\001' && p[3] == '\002') || (p[2] == '\003' && p[3] == '\004') ||
		    (p[2] == '\005' && p[3] == '\006') || (p[2] == '\006' && p[3] == '\006') ||
		    (p[2] == '\007' && p[3] == '\010') || (p[2] == '0' && p[3] == '0'))
			return 29;
	}
	return 0;
}

/* archive_read_support_format_zip.c:4212-4229 */
static int64_t dos_to_unix(uint32_t msTime)
{
	uint16_t msDate = (uint16_t)(msTime >> 16), t = (uint16_t)msTime;
	struct tm ts;
	memset(&ts, 0, sizeof(ts));
	ts.tm_year = ((msDate >> 9) & 0x7f) + 80;
	ts.tm_mon = ((msDate >> 5) & 0x0f) - 1;
	ts.tm_mday = msDate & 0x1f;
	ts.tm_hour = (t >> 11) & 0x1f;
	ts.tm_min = (t >> 5) & 0x3f;
	ts.tm_sec = (t << 1) & 0x3e;
	ts.tm_isdst = -1;
	return (int64_t)mktime(&ts);
}

static int zip_fail(struct archive_read *a, int rc, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
static int zip_fail(struct archive_read *a, int rc, const char *fmt, ...)
{
	char buf[400];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC, "%s", buf);
	return rc;
}

/* gather the whole archive (the core of this slice cannot seek) and walk its central directory */
static int zip_load(struct archive_read *a, struct zip_private *z)
{
	const char *lim = getenv("LA_ZIP_MAX_MIB");
	const size_t max_bytes = (size_t)(lim && atoi(lim) > 0 ? atoi(lim) : 16384) << 20;
	for (;;) {
		ssize_t avail;
		const void *p = __archive_read_ahead(a, 1, &avail);
		if (p == NULL) {
			if (avail < 0)
				return ARCHIVE_FATAL;
			break;
		}
		if (z->img_len + (size_t)avail > max_bytes)
			return zip_fail(a, ARCHIVE_FATAL, "ZIP archive larger than LA_ZIP_MAX_MIB (%zu MiB): this reader holds the whole archive in memory", max_bytes >> 20);
		if (z->img_len + (size_t)avail > z->img_cap) {
			size_t nc = z->img_cap ? z->img_cap : (1u << 20);
			while (nc < z->img_len + (size_t)avail) nc *= 2;
			uint8_t *nb = realloc(z->img, nc);
			if (!nb) { archive_set_error(&a->archive, ENOMEM, "Can't allocate ZIP data"); return ARCHIVE_FATAL; }
			z->img = nb; z->img_cap = nc;
		}
		memcpy(z->img + z->img_len, p, (size_t)avail);
		z->img_len += (size_t)avail;
		__archive_read_consume(a, avail);
	}
	const uint8_t *img = z->img;
	const size_t len = z->img_len;
	if (len < 22)
		return zip_fail(a, ARCHIVE_FATAL, "Truncated ZIP file header");
	/* end-of-central-directory record: the last "PK\5\6" whose comment reaches the end */
	size_t eocd = (size_t)-1;
	for (size_t back = 22; back <= len && back <= 22 + 65535; back++) {
		const uint8_t *q = img + len - back;
		if (q[0] == 'P' && q[1] == 'K' && q[2] == 5 && q[3] == 6 && (size_t)le16(q + 20) + 22 <= back) {
			eocd = len - back;
			break;
		}
	}
	if (eocd == (size_t)-1)
		return zip_fail(a, ARCHIVE_FATAL, "ZIP central directory not found (truncated archive?)");
	uint64_t total = le16(img + eocd + 10), cd_size = le32(img + eocd + 12), cd_off = le32(img + eocd + 16);
	uint64_t cd_end = eocd;
	if (eocd >= 20 && memcmp(img + eocd - 20, "PK\006\007", 4) == 0) {
		/* Zip64 locator -> Zip64 end record */
		uint64_t r = le64(img + eocd - 20 + 8);
		if (r < len && len - r >= 56 && memcmp(img + r, "PK\006\006", 4) == 0) {
			total = le64(img + r + 32);
			cd_size = le64(img + r + 40);
			cd_off = le64(img + r + 48);
			cd_end = r;
		}
	}
	if (cd_size > cd_end)
		return zip_fail(a, ARCHIVE_FATAL, "Damaged ZIP central directory");
	/* bytes in front of the archive shift every stored offset (:3960-3990 "correction") */
	const uint64_t cd_pos = cd_end - cd_size;
	const int64_t shift = (int64_t)cd_pos - (int64_t)cd_off;
	if (total > cd_size / 46)	/* (every record is at least 46 bytes) */
		return zip_fail(a, ARCHIVE_FATAL, "Damaged ZIP central directory");
	z->ents = calloc((size_t)total + 1, sizeof(*z->ents));
	z->res_idx = malloc(((size_t)total + 1) * sizeof(uint32_t));
	if (!z->ents || !z->res_idx) { archive_set_error(&a->archive, ENOMEM, "Can't allocate ZIP data"); return ARCHIVE_FATAL; }
	uint64_t p = cd_pos;
	uint32_t n = 0;
	for (uint64_t i = 0; i < total; i++) {
		if (p + 46 > cd_end || memcmp(img + p, "PK\001\002", 4) != 0)
			return zip_fail(a, ARCHIVE_FATAL, "Damaged ZIP central directory");
		const uint8_t *h = img + p;
		struct zip_entry *e = &z->ents[n];
		e->system = h[5];
		e->version = h[6];
		e->flags = le16(h + 8);
		e->method = le16(h + 10);
		e->mtime = dos_to_unix(le32(h + 12));
		e->crc = le32(h + 16);
		e->csize = le32(h + 20);
		e->usize = le32(h + 24);
		const uint32_t nlen = le16(h + 28), elen = le16(h + 30), clen = le16(h + 32);
		const uint32_t eattr = le32(h + 38);
		e->lho = le32(h + 42);
		if (p + 46 + nlen + elen + clen > cd_end)
			return zip_fail(a, ARCHIVE_FATAL, "Damaged ZIP central directory");
		e->name_off = (uint64_t)(p + 46);
		e->name_len = (uint16_t)nlen;
		/* Zip64 extended information (0x0001): the 8-byte forms of whichever fields read 0xffffffff, in order */
		const uint8_t *x = h + 46 + nlen, *xe = x + elen;
		while (x + 4 <= xe) {
			const uint16_t id = le16(x), sz = le16(x + 2);
			if (x + 4 + sz > xe)
				break;
			if (id == 0x0001) {
				const uint8_t *f = x + 4, *fe = f + sz;
				if (e->usize == 0xFFFFFFFFu && f + 8 <= fe) { e->usize = le64(f); f += 8; }
				if (e->csize == 0xFFFFFFFFu && f + 8 <= fe) { e->csize = le64(f); f += 8; }
				if (e->lho == 0xFFFFFFFFu && f + 8 <= fe) { e->lho = le64(f); f += 8; }
			}
			x += 4 + sz;
		}
		e->lho = (uint64_t)((int64_t)e->lho + shift);
		if (e->lho > len || len - e->lho < 30 || memcmp(img + e->lho, "PK\003\004", 4) != 0)
			return zip_fail(a, ARCHIVE_FATAL, "Damaged ZIP archive: no local file header where the central directory points");
		e->data_off = e->lho + 30 + le16(img + e->lho + 26) + le16(img + e->lho + 28);
		const int is_dir = nlen > 0 && img[e->name_off + nlen - 1] == '/';
		if (e->system == 3 && (eattr >> 16) != 0)
			e->mode = eattr >> 16;
		else
			e->mode = is_dir ? (AE_IFDIR | 0775) : (AE_IFREG | 0664);
		if ((e->mode & AE_IFMT) == 0)
			e->mode |= is_dir ? AE_IFDIR : AE_IFREG;
		p += 46 + nlen + elen + clen;
		n++;
	}
	z->n = n;
	z->cur = 0;
	z->b_lo = z->b_hi = 0;
	z->loaded = 1;
	return ARCHIVE_OK;
}

static int zip_read_header(struct archive_read *a, struct archive_entry *entry)
{
	struct zip_private *z = a->format->data;
	if (!z->loaded) {
		int r = zip_load(a, z);
		if (r != ARCHIVE_OK)
			return r;
	}
	const uint32_t next = z->started ? z->cur + 1 : 0;
	a->archive.archive_format = ARCHIVE_FORMAT_ZIP;
	if (next >= z->n) {
		z->data_done = 1;
		if (a->archive.archive_format_name == NULL)
			a->archive.archive_format_name = "ZIP";
		return ARCHIVE_EOF;
	}
	z->started = 1;
	z->cur = next;
	z->data_done = 0;
	const struct zip_entry *e = &z->ents[next];
	size_t nl = e->name_len < sizeof(entry->pathname) - 1 ? e->name_len : sizeof(entry->pathname) - 1;
	memcpy(entry->pathname, z->img + e->name_off, nl);
	entry->pathname[nl] = 0;
	entry->mtime = e->mtime; entry->mtime_set = 1;
	entry->size = (int64_t)e->usize; entry->size_set = 1;
	entry->filetype = e->mode & AE_IFMT;
	entry->mode = e->mode & 07777;
	a->archive.archive_format = ARCHIVE_FORMAT_ZIP;
	snprintf(z->fmt_name, sizeof(z->fmt_name), "ZIP %d.%d (%s)", e->version / 10, e->version % 10,
	    e->method == 0 ? "uncompressed" : e->method == 8 ? "deflation" : "unsupported method");
	a->archive.archive_format_name = z->fmt_name;
	return ARCHIVE_OK;
}

/* decode the batch of entries that starts at `first` */
static int zip_run_batch(struct archive_read *a, struct zip_private *z, uint32_t first)
{
	if (!z->gpu) {
		const char *dev = getenv("LA_GPU_DEVICE");
		if (la_gpu_open(dev ? atoi(dev) : 0, &z->gpu) != LA_OK)
			return zip_fail(a, ARCHIVE_FATAL, "ZIP reader: no usable MI355X device (this build has no CPU inflate)");
		if (la_gpu_malloc(z->gpu, &z->d_img, z->img_len + 64) != LA_OK ||
		    la_gpu_memcpy_h2d(z->gpu, z->d_img, z->img, z->img_len) != LA_OK)
			return zip_fail(a, ARCHIVE_FATAL, "ZIP reader: device allocation failed: %s", la_gpu_last_error(z->gpu));
	}
	/* the batch: consecutive entries until the decoded bytes or the entry count reach their bounds */
	uint64_t out = 0;
	uint32_t hi = first, nm = 0;
	while (hi < z->n && hi - first < ZIP_BATCH_ENTRIES) {
		const struct zip_entry *e = &z->ents[hi];
		const int inflate = e->method == 8 && !(e->flags & (ZIP_ENCRYPTED | ZIP_STRONG_ENCRYPTED)) &&
		    e->usize < 0x80000000ull && e->csize <= 0xFFFFFFFFull && e->csize <= z->img_len && e->data_off <= z->img_len - e->csize;
		if (inflate) {
			if (nm && out + e->usize > ZIP_BATCH_OUT)
				break;
			out += (e->usize + 15) & ~15ull;
			nm++;
		}
		hi++;
	}
	la_gz_member *mem = malloc((size_t)(nm ? nm : 1) * sizeof(*mem));
	free(z->res);
	z->res = malloc((size_t)(nm ? nm : 1) * sizeof(*z->res));
	if (!mem || !z->res) { free(mem); archive_set_error(&a->archive, ENOMEM, "Can't allocate ZIP data"); return ARCHIVE_FATAL; }
	uint64_t o = 0;
	uint32_t k = 0;
	for (uint32_t i = first; i < hi; i++) {
		struct zip_entry *e = &z->ents[i];
		z->res_idx[i] = 0xFFFFFFFFu;
		const int inflate = e->method == 8 && !(e->flags & (ZIP_ENCRYPTED | ZIP_STRONG_ENCRYPTED)) &&
		    e->usize < 0x80000000ull && e->csize <= 0xFFFFFFFFull && e->csize <= z->img_len && e->data_off <= z->img_len - e->csize;
		if (!inflate)
			continue;
		mem[k].src_off = e->data_off;
		mem[k].src_len = (uint32_t)e->csize;
		mem[k].dst_cap = (uint32_t)e->usize;
		mem[k].dst_off = o;
		e->slab_off = o;
		z->res_idx[i] = k++;
		o += (e->usize + 15) & ~15ull;
	}
	z->b_lo = first; z->b_hi = hi;
	int rc = ARCHIVE_OK;
	if (nm) {
		void *d_mem = NULL, *d_res = NULL, *d_dst = NULL;
		if (z->slab_cap < o + 64) {
			if (z->slab) la_gpu_free_host(z->gpu, z->slab);
			void *hp = NULL;
			z->slab = NULL; z->slab_cap = 0;
			if (la_gpu_malloc_host(z->gpu, &hp, o + 64) != LA_OK) { free(mem); return zip_fail(a, ARCHIVE_FATAL, "ZIP reader: pinned slab allocation failed"); }
			z->slab = hp; z->slab_cap = o + 64;
		}
		la_gz_batch bt;
		memset(&bt, 0, sizeof(bt));
		if (la_gpu_malloc(z->gpu, &d_mem, (size_t)nm * sizeof(*mem)) != LA_OK ||
		    la_gpu_malloc(z->gpu, &d_res, (size_t)nm * sizeof(la_gz_result)) != LA_OK ||
		    la_gpu_malloc(z->gpu, &d_dst, o + 64) != LA_OK ||
		    la_gpu_memcpy_h2d(z->gpu, d_mem, mem, (size_t)nm * sizeof(*mem)) != LA_OK) {
			rc = zip_fail(a, ARCHIVE_FATAL, "ZIP reader: device allocation failed: %s", la_gpu_last_error(z->gpu));
		} else {
			bt.d_src = z->d_img; bt.src_bytes = z->img_len;
			bt.d_members = d_mem; bt.n_members = nm;
			bt.d_dst = d_dst; bt.dst_cap = o;
			bt.d_results = d_res;
			bt.options = LA_GZ_OPT_RAW;	/* raw deflate bodies: no gzip trailer; the CRC32 comes back in the results */
			if (la_gpu_gzip_decode(z->gpu, &bt) != LA_OK ||
			    la_gpu_memcpy_d2h(z->gpu, z->res, d_res, (size_t)nm * sizeof(la_gz_result)) != LA_OK ||
			    la_gpu_memcpy_d2h(z->gpu, z->slab, d_dst, o) != LA_OK ||
			    la_gpu_sync(z->gpu) != LA_OK)
				rc = zip_fail(a, ARCHIVE_FATAL, "ZIP reader: device decode failed: %s", la_gpu_last_error(z->gpu));
		}
		if (d_mem) la_gpu_free(z->gpu, d_mem);
		if (d_res) la_gpu_free(z->gpu, d_res);
		if (d_dst) la_gpu_free(z->gpu, d_dst);
	}
	free(mem);
	return rc;
}

static const char *method_name(int m)
{
	switch (m) {	/* archive_read_support_format_zip.c:425-450 */
	case 1: return "shrinking"; case 6: return "imploded"; case 9: return "deflation-64-bit";
	case 12: return "bzip"; case 14: return "lzma"; case 93: return "zstd"; case 95: return "xz";
	case 98: return "ppmd-1"; case 99: return "winzip-aes";
	default: return "??";
	}
}

static int zip_read_data(struct archive_read *a, const void **buff, size_t *size, int64_t *offset)
{
	struct zip_private *z = a->format->data;
	*buff = NULL; *size = 0; *offset = 0;
	if (!z->loaded || !z->started || z->cur >= z->n)
		return ARCHIVE_EOF;
	const struct zip_entry *e = &z->ents[z->cur];
	if (z->data_done) {
		*offset = (int64_t)e->usize;
		return ARCHIVE_EOF;
	}
	z->data_done = 1;
	if (e->flags & (ZIP_ENCRYPTED | ZIP_STRONG_ENCRYPTED))
		return zip_fail(a, ARCHIVE_FAILED, "Encrypted file is unsupported");
	if (e->method != 0 && e->method != 8)
		return zip_fail(a, ARCHIVE_FAILED, "Unsupported ZIP compression method (%d: %s)", e->method, method_name(e->method));
	if (e->csize > z->img_len || e->data_off > z->img_len - e->csize)
		return zip_fail(a, ARCHIVE_FATAL, "Truncated ZIP file data");
	if (e->method == 0) {
		/* stored: straight out of the gathered image; its check value like any other entry's (:3155-3171) */
		const unsigned long c = la_crc32(0, z->img + e->data_off, (size_t)e->csize);
		if (e->usize != e->csize)
			return zip_fail(a, ARCHIVE_FAILED, "ZIP uncompressed data is wrong size (read %jd, expected %jd)\n",
			    (intmax_t)e->csize, (intmax_t)e->usize);
		if ((uint32_t)c != e->crc)
			return zip_fail(a, ARCHIVE_FAILED, "ZIP bad CRC: 0x%lx should be 0x%lx", c, (unsigned long)e->crc);
		*buff = z->img + e->data_off;
		*size = (size_t)e->csize;
		return e->csize ? ARCHIVE_OK : ARCHIVE_EOF;
	}
	if (e->usize >= 0x80000000ull || e->csize > 0xFFFFFFFFull)
		return zip_fail(a, ARCHIVE_FAILED, "ZIP entry too large for the GPU data plane (2 GiB decoded per entry)");
	if (z->cur < z->b_lo || z->cur >= z->b_hi) {
		int r = zip_run_batch(a, z, z->cur);
		if (r != ARCHIVE_OK)
			return r;
	}
	const la_gz_result *r = &z->res[z->res_idx[z->cur]];
	if (r->status == LA_ST_GZ_OUT_FULL)
		return zip_fail(a, ARCHIVE_FAILED, "ZIP uncompressed data is wrong size (read more than %jd, expected %jd)\n",
		    (intmax_t)e->usize, (intmax_t)e->usize);
	if (r->status == LA_ST_GZ_TRUNCATED)
		return zip_fail(a, ARCHIVE_FATAL, "Truncated ZIP file body");
	if (r->status != LA_ST_OK)
		return zip_fail(a, ARCHIVE_FATAL, "ZIP decompression failed (%d)", -3 /* Z_DATA_ERROR */);
	if (r->crc32 != e->crc)
		return zip_fail(a, ARCHIVE_FAILED, "ZIP bad CRC: 0x%lx should be 0x%lx", (unsigned long)r->crc32, (unsigned long)e->crc);
	if ((uint64_t)r->consumed != e->csize)
		return zip_fail(a, ARCHIVE_FAILED, "ZIP compressed data is wrong size (read %jd, expected %jd)",
		    (intmax_t)r->consumed, (intmax_t)e->csize);
	if ((uint64_t)r->out_len != (e->usize & 0xFFFFFFFFull))
		return zip_fail(a, ARCHIVE_FAILED, "ZIP uncompressed data is wrong size (read %jd, expected %jd)\n",
		    (intmax_t)r->out_len, (intmax_t)e->usize);
	*buff = z->slab + e->slab_off;
	*size = r->out_len;
	return r->out_len ? ARCHIVE_OK : ARCHIVE_EOF;
}

static int zip_skip(struct archive_read *a)
{
	struct zip_private *z = a->format->data;
	z->data_done = 1;	/* entries are addressed through the directory: nothing to read past */
	return ARCHIVE_OK;
}

static int zip_cleanup(struct archive_read *a)
{
	struct zip_private *z = a->format->data;
	if (z) {
		if (z->gpu) {
			if (z->slab) la_gpu_free_host(z->gpu, z->slab);
			if (z->d_img) la_gpu_free(z->gpu, z->d_img);
			la_gpu_close(z->gpu);
		}
		free(z->img); free(z->ents); free(z->res); free(z->res_idx);
		free(z);
	}
	a->format->data = NULL;
	return ARCHIVE_OK;
}

int archive_read_support_format_zip(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	for (int i = 0; i < 4; i++)
		if (a->formats[i].bid == zip_bid)
			return ARCHIVE_OK;
	struct zip_private *z = calloc(1, sizeof(*z));
	if (z == NULL) {
		archive_set_error(_a, ENOMEM, "Can't allocate zip data");
		return ARCHIVE_FATAL;
	}
	struct archive_format_descriptor d = { z, "zip", zip_bid, zip_read_header, zip_read_data, zip_cleanup, zip_skip };
	if (__archive_read_register_format(a, d) != ARCHIVE_OK) {
		free(z);
		return ARCHIVE_FATAL;	/* (the core has set the error: no free format slot) */
	}
	return ARCHIVE_OK;
}
