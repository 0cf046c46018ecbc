/*
 * Minimal ustar walker over the decoded stream: enough of
 * archive_read_support_format_tar.c to make "tar.gz / tar.lz4 -> archive_read_next_header /
 * archive_read_data_block" a real loop on top of the device filters (BASELINE.json configs[3]).
 *
 * What it follows (file:line of the reference):
 *   bid                    archive_read_support_format_tar.c:367-430   (10 for an end mark, 48 checksum,
 *                                                                      +56 ustar magic, +2 typeflag; 0 on a bad number field)
 *   number fields          :339-365 (validate), :3388-3429 (octal, leading blanks, '-', clamp), base-256 marker
 *   header checksum        :991-1046 (unsigned sum, then the signed-char variant)
 *   header loop            :714-810  (EOF at a record boundary = end, short record = "Truncated tar archive
 *                                    detected while reading next header", one or two zero records = end,
 *                                    bad checksum = ARCHIVE_RETRY "Damaged tar archive (bad header checksum)")
 *   ustar fields           :1769-1843 (prefix + '/' + name, padding 0x1ff & -size), :1334-1590 (type flag table)
 *   trailing '/' = dir     :580-600
 *   data                   :604-668  (slices of whatever ahead(1) exposes, clipped to the entry; then the
 *                                    padding; "Truncated tar archive detected while reading data")
 *   skip                   :670-691
 *
 *   GNU headers            :2927-3023 (magic "ustar  \0": no prefix field), 'L' long name / 'K' long link :1172-1221,
 *                          :1251-1316 (body -> string, "Truncated archive detected while reading metadata"),
 *                          'V' volume header :1227-1245
 *   pax 'x' / 'X' / 'g'    :1846-2100 (records "<len> <key>=<value>\n"); of the keys only path, size and mtime
 *                          reach what this slice exposes; 'g' is read and ignored
 *   sequence rules         :748-986 (one of each special header per entry, EOF or a bad checksum after a special
 *                          header is fatal: "Damaged tar archive (end-of-archive within a sequence of headers)")
 *
 * Out of this slice on purpose: sparse entries (GNU 'S', GNU.sparse.* pax keys), Solaris ACLs ('A'), mac metadata,
 * character-set conversion, names longer than the entry's 1023-byte buffer.  Such input is refused with a message
 * saying so rather than half-read.
 */
#include <errno.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "la_read_private.h"

struct ustar_hdr {			/* archive_read_support_format_tar.c:85-104 */
	char name[100], mode[8], uid[8], gid[8], size[12], mtime[12], checksum[8], typeflag[1];
	char linkname[100], magic[6], version[2], uname[32], gname[32], rdevmajor[8], rdevminor[8];
	char prefix[155], pad[12];
};

struct tar_info {
	int64_t entry_bytes_remaining;
	int64_t entry_bytes_unconsumed;
	int64_t entry_padding;
	int64_t entry_offset;
	int64_t disk_size;
	/* what the special headers in front of a regular header said (reset per entry) */
	int     have_path, have_size, have_mtime;
	char    path[1024];
	int64_t pax_size, pax_mtime;
};

static int block_is_null(const char *h)
{
	for (int i = 0; i < 512; i++)
		if (h[i])
			return 0;
	return 1;
}

static int64_t atol_base_n(const char *p, size_t n, int base)
{
	int64_t l = 0, maxval = INT64_MAX, limit = INT64_MAX / base, last = INT64_MAX % base;
	int sign = 1;
	while (n && (*p == ' ' || *p == '\t')) { p++; n--; }
	if (n && *p == '-') {
		sign = -1; p++; n--;
		maxval = INT64_MIN;
		limit = -(INT64_MIN / base);
		last = -(INT64_MIN % base);
	}
	while (n) {
		int d = *p - '0';
		if (d < 0 || d >= base)
			break;
		if (l > limit || (l == limit && d >= last))
			return maxval;
		l = l * base + d;
		p++; n--;
	}
	return sign < 0 ? -l : l;
}

/* base-256 (GNU / star, :3454-3494): big-endian two's complement whose first byte carries the marker in
 * bit 7 and the sign in bit 6; anything that does not fit 64 bits clamps */
static int64_t atol256(const char *f, size_t n)
{
	const unsigned char *p = (const unsigned char *)f;
	const int negative = (p[0] & 0x40) != 0;
	const unsigned char fill = negative ? 0xff : 0x00;
	const int64_t clamp = negative ? INT64_MIN : INT64_MAX;
	unsigned char cur = negative ? (unsigned char)(p[0] | 0x80) : (unsigned char)(p[0] & 0x7f);
	size_t i = 0;
	for (; n - i > 8; i++) {		/* bytes above the low eight: sign extension only */
		if (cur != fill)
			return clamp;
		cur = p[i + 1];
	}
	if ((cur ^ fill) & 0x80)
		return clamp;
	uint64_t v = negative ? ~(uint64_t)0 : 0;
	v = (v << 8) | cur;
	for (i++; i < n; i++)
		v = (v << 8) | p[i];
	return (int64_t)v;
}

static int64_t tar_atol(const char *p, size_t n)
{
	return (*p & 0x80) ? atol256(p, n) : atol_base_n(p, n, 8);
}

static int number_field_ok(const char *f, size_t n)
{
	unsigned char m = (unsigned char)f[0];
	size_t i = 0;
	if (m == 128 || m == 255 || m == 0)
		return 1;
	while (i < n && f[i] == ' ') i++;
	while (i < n && f[i] >= '0' && f[i] <= '7') i++;
	while (i < n) {
		if (f[i] != ' ' && f[i] != 0)
			return 0;
		i++;
	}
	return 1;
}

static int checksum_ok(const char *h)
{
	const struct ustar_hdr *u = (const struct ustar_hdr *)h;
	const unsigned char *b = (const unsigned char *)h;
	for (size_t i = 0; i < sizeof(u->checksum); i++) {
		char c = u->checksum[i];
		if (c != ' ' && c != 0 && (c < '0' || c > '7'))
			return 0;
	}
	int sum = (int)tar_atol(u->checksum, sizeof(u->checksum));
	int un = 0, sg = 0;
	for (int i = 0; i < 512; i++) {
		if (i >= 148 && i < 156) { un += 32; sg += 32; }
		else { un += b[i]; sg += (signed char)b[i]; }
	}
	return sum == un || sum == sg;
}

static int tar_bid(struct archive_read *a, int best_bid)
{
	(void)best_bid;
	const char *h = __archive_read_ahead(a, 512, NULL);
	if (h == NULL)
		return -1;
	if (h[0] == 0 && block_is_null(h))
		return 10;
	if (!checksum_ok(h))
		return 0;
	int bid = 48;
	const struct ustar_hdr *u = (const struct ustar_hdr *)h;
	if (memcmp(u->magic, "ustar\0", 6) == 0 && memcmp(u->version, "00", 2) == 0)
		bid += 56;
	if (memcmp(u->magic, "ustar ", 6) == 0 && memcmp(u->version, " \0", 2) == 0)
		bid += 56;
	char t = u->typeflag[0];
	if (t != 0 && !(t >= '0' && t <= '9') && !(t >= 'A' && t <= 'Z') && !(t >= 'a' && t <= 'z'))
		return 0;
	bid += 2;
	if (!number_field_ok(u->mode, sizeof(u->mode)) || !number_field_ok(u->uid, sizeof(u->uid)) ||
	    !number_field_ok(u->gid, sizeof(u->gid)) || !number_field_ok(u->mtime, sizeof(u->mtime)) ||
	    !number_field_ok(u->size, sizeof(u->size)) || !number_field_ok(u->rdevmajor, sizeof(u->rdevmajor)) ||
	    !number_field_ok(u->rdevminor, sizeof(u->rdevminor)))
		bid = 0;
	return bid;
}

static void flush_unconsumed(struct archive_read *a, int64_t *unconsumed)
{
	if (*unconsumed) {
		__archive_read_consume(a, *unconsumed);
		*unconsumed = 0;
	}
}

static size_t field_len(const char *f, size_t n)
{
	size_t i = 0;
	while (i < n && f[i]) i++;
	return i;
}

static const int64_t entry_limit = 0xfffffffffffffffLL;	/* :249 */

static int header_fields(struct archive_read *a, struct tar_info *tar, struct archive_entry *entry,
    const struct ustar_hdr *u, int is_ustar)
{
	char path[100 + 155 + 2];
	size_t n = 0;
	if (is_ustar && u->prefix[0]) {
		n = field_len(u->prefix, sizeof(u->prefix));
		memcpy(path, u->prefix, n);
		if (path[n - 1] != '/')
			path[n++] = '/';
	}
	size_t k = field_len(u->name, sizeof(u->name));
	memcpy(path + n, u->name, k);
	path[n + k] = 0;
	archive_entry_set_pathname(entry, tar->have_path ? tar->path : path);

	int64_t mode = tar_atol(u->mode, sizeof(u->mode));
	entry->mode = (unsigned)mode & 07777;
	entry->mtime = tar->have_mtime ? tar->pax_mtime : tar_atol(u->mtime, sizeof(u->mtime));
	entry->mtime_set = 1;

	tar->disk_size = tar->have_size ? tar->pax_size : tar_atol(u->size, sizeof(u->size));	/* :1378-1383, :1403-1408 */
	if (tar->disk_size < 0) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC, "Tar entry has negative file size");
		return ARCHIVE_FATAL;
	}
	if (tar->disk_size > entry_limit) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC, "Tar entry size overflow");
		return ARCHIVE_FATAL;
	}
	entry->size = tar->disk_size;
	entry->size_set = 1;
	tar->entry_bytes_remaining = tar->disk_size;

	switch (u->typeflag[0]) {
	case '1':	/* hard link: a ustar body is ignored when a valid header follows; this slice keeps size 0 only
			 * for the zero-size case and otherwise treats the body as data like pax does (:1459-1494) */
		entry->filetype = entry->size > 0 ? AE_IFREG : 0;
		break;
	case '2': entry->filetype = AE_IFLNK; entry->size = 0; tar->entry_bytes_remaining = 0; break;
	case '3': entry->filetype = AE_IFCHR; entry->size = 0; tar->entry_bytes_remaining = 0; break;
	case '4': entry->filetype = AE_IFBLK; entry->size = 0; tar->entry_bytes_remaining = 0; break;
	case '5': entry->filetype = AE_IFDIR; entry->size = 0; tar->entry_bytes_remaining = 0; break;
	case '6': entry->filetype = AE_IFIFO; entry->size = 0; tar->entry_bytes_remaining = 0; break;
	case 'D': entry->filetype = AE_IFDIR; break;
	default:  entry->filetype = AE_IFREG; break;
	}
	tar->entry_padding = 0x1ff & (-tar->entry_bytes_remaining);
	return ARCHIVE_OK;
}

/* body of a special header -> buf (NUL terminated); the body and its padding join *unconsumed (:1251-1316) */
static int body_to_string(struct archive_read *a, const struct ustar_hdr *u, char *buf, size_t cap, size_t *len,
    int64_t *unconsumed)
{
	int64_t size = tar_atol(u->size, sizeof(u->size));
	if (size < 0 || size > entry_limit) {
		archive_set_error(&a->archive, EINVAL, "Special header has invalid size: %lld", (long long)size);
		return ARCHIVE_FATAL;
	}
	if (size > 1048576) {	/* pathname_limit :244 */
		int64_t to_consume = (size + 511) & ~(int64_t)511;
		flush_unconsumed(a, unconsumed);
		if (to_consume != __archive_read_consume(a, to_consume))
			return ARCHIVE_FATAL;
		archive_set_error(&a->archive, EINVAL, "Special header too large: %lld > 1MiB", (long long)size);
		*len = 0;
		buf[0] = 0;
		return ARCHIVE_WARN;
	}
	flush_unconsumed(a, unconsumed);
	const char *src = __archive_read_ahead(a, (size_t)size, NULL);
	if (src == NULL && size > 0) {
		archive_set_error(&a->archive, EINVAL, "Truncated archive detected while reading metadata");
		return ARCHIVE_FATAL;
	}
	if ((size_t)size >= cap) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT,
		    "tar metadata of %lld bytes is outside this slice (at most %zu)", (long long)size, cap - 1);
		return ARCHIVE_FATAL;
	}
	if (size)
		memcpy(buf, src, (size_t)size);
	buf[size] = 0;
	*len = (size_t)size;
	*unconsumed += size + (0x1ff & (-size));
	return ARCHIVE_OK;
}

/* pax records "<decimal length> <key>=<value>\n" (:1846-2100); keys this slice exposes: path, size, mtime */
static int pax_records(struct archive_read *a, struct tar_info *tar, const char *p, size_t n, int global)
{
	while (n > 0) {
		size_t l = 0, i = 0;
		while (i < n && p[i] >= '0' && p[i] <= '9' && i < 10)
			l = l * 10 + (size_t)(p[i++] - '0');
		if (i == 0 || i >= n || p[i] != ' ' || l <= i + 1 || l > n || p[l - 1] != '\n') {
			archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC, "Ignoring malformed pax extended attributes");
			return ARCHIVE_WARN;
		}
		const char *key = p + i + 1, *end = p + l - 1;
		const char *eq = memchr(key, '=', (size_t)(end - key));
		if (eq == NULL) {
			archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC, "Invalid pax extended attributes");
			return ARCHIVE_WARN;
		}
		const size_t kl = (size_t)(eq - key), vl = (size_t)(end - eq - 1);
		const char *v = eq + 1;
		if (!global) {
			if (kl >= 11 && memcmp(key, "GNU.sparse.", 11) == 0) {
				archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT, "sparse tar entries are outside this slice");
				return ARCHIVE_FATAL;
			}
			if (kl == 4 && memcmp(key, "path", 4) == 0) {
				if (vl >= sizeof(tar->path)) {
					archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT,
					    "pathname of %zu bytes is outside this slice (at most %zu)", vl, sizeof(tar->path) - 1);
					return ARCHIVE_FATAL;
				}
				memcpy(tar->path, v, vl);
				tar->path[vl] = 0;
				tar->have_path = 1;
			} else if (kl == 4 && memcmp(key, "size", 4) == 0) {
				tar->pax_size = atol_base_n(v, vl, 10);
				tar->have_size = 1;
			} else if (kl == 5 && memcmp(key, "mtime", 5) == 0) {
				tar->pax_mtime = atol_base_n(v, vl, 10);	/* whole seconds; the fraction is not exposed */
				tar->have_mtime = 1;
			}
		}
		p += l;
		n -= l;
	}
	return ARCHIVE_OK;
}

static int err_combine(int a, int b) { return a < b ? a : b; }

static int tar_read_header(struct archive_read *a, struct archive_entry *entry)
{
	struct tar_info *tar = a->format->data;
	int64_t unconsumed = 0;
	ssize_t bytes;
	const char *h;
	int err = ARCHIVE_OK, eof_fatal = 0;
	unsigned seen = 0;	/* 1 'g', 2 'K', 4 'L', 8 'V', 16 'x'/'X' */
	char *meta = NULL;	/* body of a special header */
	const size_t meta_cap = 65536;

	tar->entry_offset = 0;
	tar->have_path = tar->have_size = tar->have_mtime = 0;
	if (a->archive.archive_format_name == NULL || (a->archive.archive_format & 0xff0000) != ARCHIVE_FORMAT_TAR) {
		a->archive.archive_format = ARCHIVE_FORMAT_TAR;
		a->archive.archive_format_name = "tar";
	}

	for (;;) {
		flush_unconsumed(a, &unconsumed);
		h = __archive_read_ahead(a, 512, &bytes);
		if (bytes == 0) {
			free(meta);
			if (eof_fatal) {
				archive_set_error(&a->archive, EINVAL,
				    "Damaged tar archive (end-of-archive within a sequence of headers)");
				return ARCHIVE_FATAL;
			}
			return ARCHIVE_EOF;
		}
		if (h == NULL) {
			free(meta);
			archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT,
			    "Truncated tar archive detected while reading next header");
			return ARCHIVE_FATAL;
		}
		unconsumed = 512;
		if (h[0] == 0 && block_is_null(h)) {
			flush_unconsumed(a, &unconsumed);
			h = __archive_read_ahead(a, 512, NULL);
			if (h != NULL && h[0] == 0 && block_is_null(h))
				__archive_read_consume(a, 512);
			archive_clear_error(&a->archive);
			free(meta);
			return ARCHIVE_EOF;
		}
		if (!checksum_ok(h)) {
			flush_unconsumed(a, &unconsumed);
			free(meta);
			archive_set_error(&a->archive, EINVAL, "Damaged tar archive (bad header checksum)");
			return eof_fatal ? ARCHIVE_FATAL : ARCHIVE_RETRY;
		}

		const struct ustar_hdr *u = (const struct ustar_hdr *)h;
		const char t = u->typeflag[0];
		int r = ARCHIVE_OK;
		if (t == 'g' || t == 'K' || t == 'L' || t == 'V' || t == 'X' || t == 'x') {
			const unsigned bit = t == 'g' ? 1u : t == 'K' ? 2u : t == 'L' ? 4u : t == 'V' ? 8u : 16u;
			if (seen & bit) {
				if (t == 'g' || t == 'x' || t == 'X') {
					archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC,
					    t == 'g' ? "Redundant 'g' header" : t == 'x' ? "Redundant 'x' header" : "Redundant 'X'/'x' header");
					free(meta);
					return ARCHIVE_FATAL;
				}
				archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC,
				    t == 'K' ? "Damaged archive: Redundant 'K' headers may cause linknames to be incorrect"
				    : t == 'L' ? "Damaged archive: Redundant 'L' headers may cause filenames to be incorrect"
				    : "Redundant 'V' header");
				err = err_combine(err, ARCHIVE_WARN);
			}
			seen |= bit;
			if (t == 'K' || t == 'L' || t == 'V') {
				a->archive.archive_format = ARCHIVE_FORMAT_TAR_GNUTAR;
				a->archive.archive_format_name = "GNU tar format";
			} else {
				a->archive.archive_format = ARCHIVE_FORMAT_TAR_PAX_INTERCHANGE;
				a->archive.archive_format_name = t == 'X' ? "POSIX pax interchange format (Sun variant)"
				    : "POSIX pax interchange format";
			}
			if (t == 'V') {		/* volume label: the body is skipped (:1227-1245) */
				int64_t size = tar_atol(u->size, sizeof(u->size));
				if (size < 0 || size > 1048576) {
					free(meta);
					return ARCHIVE_FATAL;
				}
				unconsumed += (size + 511) & ~(int64_t)511;
			} else {
				size_t len = 0;
				if (meta == NULL && (meta = malloc(meta_cap)) == NULL) {
					archive_set_error(&a->archive, ENOMEM, "No memory");
					return ARCHIVE_FATAL;
				}
				const struct ustar_hdr keep = *u;	/* the look-ahead below may move the window */
				r = body_to_string(a, &keep, meta, meta_cap, &len, &unconsumed);
				if (r == ARCHIVE_OK && t == 'L') {
					len = strlen(meta);		/* GNU tar writes the terminating NUL into the body */
					if (len >= sizeof(tar->path)) {
						archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT,
						    "pathname of %zu bytes is outside this slice (at most %zu)", len, sizeof(tar->path) - 1);
						r = ARCHIVE_FATAL;
					} else {
						memcpy(tar->path, meta, len + 1);
						tar->have_path = 1;
					}
				} else if (r == ARCHIVE_OK && (t == 'x' || t == 'X' || t == 'g')) {
					r = pax_records(a, tar, meta, len, t == 'g');
				}
			}
			err = err_combine(err, r);
			if (err == ARCHIVE_FATAL) {
				free(meta);
				return err;
			}
			if (seen & ~(8u | 1u))	/* after anything but 'V' and 'g' a regular header has to follow (:981-984) */
				eof_fatal = 1;
			continue;
		}
		if (t == 'A') {
			free(meta);
			archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT, "Solaris tar ACL headers are outside this slice");
			return ARCHIVE_FATAL;
		}

		/* a regular header: old tar, ustar or GNU */
		if (memcmp(u->magic, "ustar  \0", 8) == 0) {
			a->archive.archive_format = ARCHIVE_FORMAT_TAR_GNUTAR;
			a->archive.archive_format_name = "GNU tar format";
			if (t == 'S' || h[386] != 0) {	/* sparse[0].offset[0] (:3007) */
				free(meta);
				archive_set_error(&a->archive, ARCHIVE_ERRNO_FILE_FORMAT, "sparse tar entries are outside this slice");
				return ARCHIVE_FATAL;
			}
			r = header_fields(a, tar, entry, u, 0);
		} else if (memcmp(u->magic, "ustar", 5) == 0) {
			if (a->archive.archive_format != ARCHIVE_FORMAT_TAR_PAX_INTERCHANGE) {
				a->archive.archive_format = ARCHIVE_FORMAT_TAR_USTAR;
				a->archive.archive_format_name = "POSIX ustar format";
			}
			r = header_fields(a, tar, entry, u, 1);
		} else {
			a->archive.archive_format = ARCHIVE_FORMAT_TAR;
			a->archive.archive_format_name = "tar (non-POSIX)";
			r = header_fields(a, tar, entry, u, 0);
		}
		err = err_combine(err, r);
		break;
	}
	free(meta);
	flush_unconsumed(a, &unconsumed);
	if (err < ARCHIVE_WARN)
		return ARCHIVE_FATAL;
	if (err == ARCHIVE_OK && entry->filetype == AE_IFREG) {
		size_t l = strlen(entry->pathname);
		if (l > 0 && entry->pathname[l - 1] == '/') {
			entry->filetype = AE_IFDIR;
			tar->entry_bytes_remaining = 0;
			tar->entry_padding = 0;
		}
	}
	return err;
}

static int tar_read_data(struct archive_read *a, const void **buff, size_t *size, int64_t *offset)
{
	struct tar_info *tar = a->format->data;
	ssize_t bytes_read;

	if (tar->entry_bytes_unconsumed) {
		__archive_read_consume(a, tar->entry_bytes_unconsumed);
		tar->entry_bytes_unconsumed = 0;
	}
	if (tar->entry_bytes_remaining == 0) {
		int64_t request = tar->entry_padding;
		if (__archive_read_consume(a, request) != request)
			return ARCHIVE_FATAL;
		tar->entry_padding = 0;
		*buff = NULL;
		*size = 0;
		*offset = tar->disk_size;
		return ARCHIVE_EOF;
	}
	*buff = __archive_read_ahead(a, 1, &bytes_read);
	if (*buff == NULL) {
		archive_set_error(&a->archive, ARCHIVE_ERRNO_MISC,
		    "Truncated tar archive detected while reading data");
		return ARCHIVE_FATAL;
	}
	if (bytes_read > tar->entry_bytes_remaining)
		bytes_read = (ssize_t)tar->entry_bytes_remaining;
	*size = (size_t)bytes_read;
	*offset = tar->entry_offset;
	tar->entry_offset += bytes_read;
	tar->entry_bytes_remaining -= bytes_read;
	tar->entry_bytes_unconsumed = bytes_read;
	return ARCHIVE_OK;
}

static int tar_skip(struct archive_read *a)
{
	struct tar_info *tar = a->format->data;
	int64_t request = tar->entry_bytes_remaining + tar->entry_padding + tar->entry_bytes_unconsumed;
	if (__archive_read_consume(a, request) != request)
		return ARCHIVE_FATAL;
	tar->entry_bytes_remaining = 0;
	tar->entry_bytes_unconsumed = 0;
	tar->entry_padding = 0;
	return ARCHIVE_OK;
}

static int tar_cleanup(struct archive_read *a)
{
	free(a->format->data);
	a->format->data = NULL;
	return ARCHIVE_OK;
}

int archive_read_support_format_tar(struct archive *_a)
{
	struct archive_read *a = (struct archive_read *)_a;
	for (int i = 0; i < 4; i++)
		if (a->formats[i].bid == tar_bid)
			return ARCHIVE_OK;
	struct tar_info *tar = calloc(1, sizeof(*tar));
	if (tar == NULL) {
		archive_set_error(_a, ENOMEM, "Can't allocate tar data");
		return ARCHIVE_FATAL;
	}
	struct archive_format_descriptor d = { tar, "tar", tar_bid, tar_read_header, tar_read_data, tar_cleanup, tar_skip };
	if (__archive_read_register_format(a, d) != ARCHIVE_OK)
		free(tar);
	return ARCHIVE_OK;
}
