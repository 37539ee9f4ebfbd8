"""Seeded corruption fuzz of the host parser + the device functions on the CPU harness (tests/emul), built with AddressSanitizer + UndefinedBehaviorSanitizer.
Every variant must decode or be rejected; any sanitizer report aborts the child process and fails the run.  Development tool (no GPU, no reference needed).

    python tools/fuzz_emul_sanitized.py [variants per file, default 60] [name substring]
"""
import ctypes as C
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# one file per kind of stream the decoder walks (round 4's last additions at the end: delta palettes, > 64 clusters, Modular channels spread over passes,
# two levels of LF frames, noise on an upsampled frame, previous-channel properties, upsampled animation layers)
NAMES = ["v256_e7", "v264x520_e7", "l200x120_e7", "va300x520_e7", "v64_hard_e7", "lra200x150_e5", "va400x300_e7_d2", "asset_animated", "j420_200x136",
         "an_modes_lossless", "u96x64_lf_frame", "ls400x300_e7", "vs400x300_e7_d1", "vu400x300_e7_d10", "vn300x200_e7",
         "lpl400x300_e7_nopatch", "lpl200x136_e7_photo", "lpl400x300_e7", "vapr400x300_e7", "vaqr520x300_e7", "vlfq600x410_e7", "vlf2_600x410_e7_d2",
         "vlf2a520x300_e7", "vnu523x267_e7_d12", "lpc200x136_e7_prev3", "lpcr200x136_e7_prev3", "an_blend_d12_e7", "an_modes_d15_e7",
         # round 5: what tools/jxl_write.py writes — splines, DCT128 / 256, custom upsampling weights, a preview frame, dequant encodings 1 - 6, 6 / 11 passes
         "w_spline_b", "w_dct_mix_a", "w_dct256", "w_up4_custom", "w_preview", "w_dequant_a", "w_dequant_b", "w_passes6", "w_passes11",
         # round 6: dequant forms rotated over the 8 x 8 tables, 24- / 20-bit integer samples, grey and hard-edged VarDCT frames, splines with the area limits
         "w_dequant_c", "l24_200x136_e7", "l20g_200x136_e3", "vhg800x600_e7_d1", "vha640x480_e7_d1", "w_spline_a", "w_spline_c"]

CHILD = r"""
import ctypes as C, random, sys, os
lib = C.CDLL(sys.argv[1]); lib.emul_last_error.restype = C.c_char_p
name, n, seed = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
d0 = open(os.path.join(sys.argv[5], "tests", "golden", name + ".jxl"), "rb").read()
rnd = random.Random(seed)
buf = C.create_string_buffer(64 << 20)
ok = bad = 0
for it in range(n):
    d = bytearray(d0)
    mode = rnd.randrange(4)
    if mode == 0:
        for _ in range(rnd.randrange(1, 4)): d[rnd.randrange(len(d))] ^= 1 << rnd.randrange(8)
    elif mode == 1:
        for _ in range(rnd.randrange(1, 8)): d[rnd.randrange(len(d))] = rnd.randrange(256)
    elif mode == 2:
        d = d[:rnd.randrange(1, len(d))]
    else:
        a = rnd.randrange(len(d)); b = min(len(d), a + rnd.randrange(1, 64)); d[a:b] = bytes(rnd.randrange(256) for _ in range(b - a))
    w, h, bits = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.emul_set_target_frame(rnd.randrange(-1, 4))
    rc = lib.emul_decode(bytes(d), len(d), 1, buf, len(buf), C.byref(w), C.byref(h), C.byref(bits))
    ok += rc == 0; bad += rc != 0
print(name, "decoded", ok, "rejected", bad)
"""


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    so = "/tmp/libjxlemul_asan.so"
    srcs = [os.path.join(ROOT, "tests", "emul", "emul.cpp"), os.path.join(ROOT, "jxl_coder_amd", "csrc", "host_parse.cpp"), os.path.join(ROOT, "jxl_coder_amd", "csrc", "host_bits.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-fPIC", "-shared",
                    "-Wno-unused-function", "-o", so] + srcs, check=True)
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    total = 0
    for i, name in enumerate(NAMES):
        if sub not in name:
            continue
        r = subprocess.run([sys.executable, "-c", CHILD, so, name, str(n), str(20260930 + i), ROOT], env=env, capture_output=True, text=True, timeout=3600)
        if r.returncode != 0:
            print("SANITIZER / CRASH in", name, "\n", r.stderr[-3000:])
            sys.exit(1)
        print(r.stdout.strip())
        total += n
    print("clean:", total, "variants")


if __name__ == "__main__":
    main()
