// tests/boundary/ref_anim_entry.cpp — C entry points for the REFERENCE'S OWN animated decoder: this file is compiled together with
// jxlcoder/src/main/cpp/interop/JxlAnimatedDecoder.cpp (unchanged, from where it lies in the reference tree) and linked against
// jxl_coder_amd/compat/libjxl.so + libjxl_threads.so (include/jxl_amd_libjxl.h): the reference's constructor walk (coalescing off, JxlDecoderSkipCurrentFrame)
// and its getFrame (JxlDecoderRewind, JxlDecoderSkipFrames, coalescing on) run against the secondary drop-in boundary and end in the HIP kernels.
// Test infrastructure; built in the build container only (tests/test_libjxl_abi.py), nothing of the reference is copied.
#include "interop/JxlAnimatedDecoder.hpp"
#include <string.h>

static void copy_msg(char *msg, size_t cap, const char *what) { if (msg && cap) { strncpy(msg, what, cap - 1); msg[cap - 1] = 0; } }

extern "C" void *refanim_open(const uint8_t *jxl, size_t size, char *msg, size_t cap) {
  try {
    std::vector<uint8_t> v(jxl, jxl + size);
    return new JxlAnimatedDecoder(v);
  } catch (std::exception &e) { copy_msg(msg, cap, e.what()); return nullptr; }
}
extern "C" void refanim_close(void *h) { delete (JxlAnimatedDecoder *)h; }
extern "C" int refanim_frames(void *h) { return ((JxlAnimatedDecoder *)h)->getNumberOfFrames(); }
extern "C" int refanim_duration(void *h, int i) { return ((JxlAnimatedDecoder *)h)->getFrameDuration(i); }
extern "C" int refanim_loops(void *h) { return (int)((JxlAnimatedDecoder *)h)->getLoopCount(); }
extern "C" void refanim_size(void *h, uint32_t *wh) { wh[0] = ((JxlAnimatedDecoder *)h)->getWidth(); wh[1] = ((JxlAnimatedDecoder *)h)->getHeight(); }
// returns 1 ok (RGBA8 in out, duration in ms, whether the reference would use the enum colour encoding), 0 on an AnimatedDecoderError (message in msg)
extern "C" int refanim_get_frame(void *h, int index, uint8_t *out, size_t out_cap, int *duration, int *prefer_encoding, char *msg, size_t cap) {
  try {
    JxlFrame f = ((JxlAnimatedDecoder *)h)->getFrame(index);
    if (f.pixels.size() > out_cap) { copy_msg(msg, cap, "buffer"); return 0; }
    memcpy(out, f.pixels.data(), f.pixels.size());
    *duration = f.duration; *prefer_encoding = f.preferColorEncoding;
    return 1;
  } catch (std::exception &e) { copy_msg(msg, cap, e.what()); return 0; }
}
