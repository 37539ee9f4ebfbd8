#!/usr/bin/env python3
"""Generate assembly trampolines: every bionic `S@LIBC` import of the prebuilt
Android x86_64 libjxl forwards to glibc's `S` (tail-jump keeps any signature)."""
import sys
PLAIN = """__cxa_atexit __cxa_finalize __memcpy_chk __memset_chk __stack_chk_fail abort atan2 cbrtf
closelog cos dl_iterate_phdr exit exp fmod free hypot hypotf ldexp ldexpf llroundf log log1p log1pf log2
log2f logf lroundf malloc memchr memcmp memcpy memmove memset modff openlog posix_memalign pow powf
pthread_cond_broadcast pthread_cond_destroy pthread_cond_signal pthread_cond_wait pthread_create
pthread_getspecific pthread_join pthread_key_create pthread_key_delete pthread_mutex_destroy pthread_mutex_lock
pthread_mutex_unlock pthread_once pthread_rwlock_rdlock pthread_rwlock_unlock pthread_rwlock_wrlock
pthread_setspecific realloc remainder sin snprintf sqrt sqrtf strcmp strlen syscall syslog vasprintf
vsnprintf wmemchr""".split()
out = ['\t.text']
for s in PLAIN:
    out += [f'\t.globl shim_{s}', f'\t.type shim_{s},@function', f'shim_{s}:', f'\tjmp {s}@PLT',
            f'\t.symver shim_{s},{s}@LIBC']
out.append('\t.section .note.GNU-stack,"",@progbits')
open(sys.argv[1], 'w').write('\n'.join(out) + '\n')
