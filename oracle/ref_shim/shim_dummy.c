/* libm.so / libdl.so stand-in sonames: one dummy symbol under version node LIBC so the
 * loader's version-need check passes; the real functions resolve through libc.so shim. */
int SHIM_DUMMY_NAME(void) { return 0; }
