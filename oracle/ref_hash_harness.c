/*
 * ref_hash_harness.c -- builds the REAL reference hash code into oracle/_ref/.
 *
 * This file contains no reference code.  It only #includes the reference's
 * header-only crc32() (libarchive/archive_crc32.h:43-84) from where it lies
 * under /root/reference and gives it an exported name; the reference's
 * libarchive/xxhash.c is compiled beside it, untouched, by oracle/Makefile
 * (config macros come from the reference's own hand-built
 * contrib/android/config/linux_host.h via its PLATFORM_CONFIG_H hook,
 * archive_platform.h:42-44).  Output goes to oracle/_ref/ only (git-ignored).
 */
#include <stddef.h>
#include "archive_crc32.h"

unsigned long ref_crc32(unsigned long crc, const void *p, size_t len);
unsigned long ref_crc32(unsigned long crc, const void *p, size_t len)
{
	return crc32(crc, p, len);
}
