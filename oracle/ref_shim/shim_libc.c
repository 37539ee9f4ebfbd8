/* bionic->glibc loader shim for the reference's prebuilt Android x86_64 libjxl
 * (reference: jxlcoder/src/main/cpp/lib/x86_64/*.so).  TEST INFRASTRUCTURE ONLY.
 * Special cases that cannot be plain tail-jumps. Exported under version node LIBC. */
#define _GNU_SOURCE
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

/* bionic LP64 sizeof(FILE) == 152; __sF[0..2] = stdin/stdout/stderr */
char shim___sF[3 * 152];
__asm__(".symver shim___sF,__sF@LIBC");

static FILE *mapf(void *f) {
  char *p = (char *)f;
  if (p >= shim___sF && p < shim___sF + sizeof(shim___sF)) {
    long i = (p - shim___sF) / 152;
    return i == 0 ? stdin : i == 1 ? stdout : stderr;
  }
  return (FILE *)f;
}
int shim_fflush(void *f) { return fflush(f ? mapf(f) : NULL); }
__asm__(".symver shim_fflush,fflush@LIBC");
int shim_fputc(int c, void *f) { return fputc(c, mapf(f)); }
__asm__(".symver shim_fputc,fputc@LIBC");
size_t shim_fwrite(const void *p, size_t s, size_t n, void *f) { return fwrite(p, s, n, mapf(f)); }
__asm__(".symver shim_fwrite,fwrite@LIBC");
int shim_vfprintf(void *f, const char *fmt, va_list ap) { return vfprintf(mapf(f), fmt, ap); }
__asm__(".symver shim_vfprintf,vfprintf@LIBC");
int shim_fprintf(void *f, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); int r = vfprintf(mapf(f), fmt, ap); va_end(ap); return r;
}
__asm__(".symver shim_fprintf,fprintf@LIBC");
int *shim___errno(void) { return &errno; }
__asm__(".symver shim___errno,__errno@LIBC");
void shim_android_set_abort_message(const char *m) { (void)m; }
__asm__(".symver shim_android_set_abort_message,android_set_abort_message@LIBC");
extern int __xpg_strerror_r(int, char *, size_t);
int shim_strerror_r(int e, char *b, size_t n) { return __xpg_strerror_r(e, b, n); }
__asm__(".symver shim_strerror_r,strerror_r@LIBC");
long shim_sysconf(int name) {
  switch (name) { /* bionic constants -> glibc */
    case 0x60: return sysconf(_SC_NPROCESSORS_CONF);
    case 0x61: return sysconf(_SC_NPROCESSORS_ONLN);
    case 0x62: return sysconf(_SC_PHYS_PAGES);
    case 0x63: return sysconf(_SC_AVPHYS_PAGES);
    case 0x27: case 0x28: return sysconf(_SC_PAGESIZE);
    default: return -1;
  }
}
__asm__(".symver shim_sysconf,sysconf@LIBC");
