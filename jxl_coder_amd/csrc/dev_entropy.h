// jxl_coder_amd/csrc/dev_entropy.h — device-side bit reader and ANS / prefix symbol reader.
// One entropy-coded stream is inherently serial (rANS state + bit position), so every stream is decoded by
// ONE lane; parallelism comes from running one wavefront per stream (135 PassGroup streams in a 4K frame,
// 34 560 in a 256-frame batch).  Replaces libjxl's ANSSymbolReader that the reference reaches through
// JxlDecoderProcessInput (jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).
//
// The functions are written against plain pointers; JXL_DEV expands to __device__ under hipcc.  The test
// suite compiles the very same header with g++ (tests/emul) to check the bitstream logic without a GPU —
// that build is test-only and never part of the shipped library.
#pragma once
#include <stdint.h>
#include "dev_types.h"
#include "dev_tables.h"

#ifdef __HIPCC__
#define JXL_DEV __device__ __forceinline__
#else
#define JXL_DEV static inline
#endif

namespace jxlamd {

// A Modular sample of a channel declared as floating point (bits total, exp_bits of exponent: float32 = 32 / 8, float16 = 16 / 5, ...): the integer IS the
// float's bit pattern; narrower formats are widened to float32 (subnormals normalised) — libjxl's int_to_float.
// integer sample of `bits` bits -> [0, 1].  libjxl multiplies by a float factor up to 22 bits and, from 23 bits on, by a DOUBLE factor before rounding to float (its
// "accurate" conversion: a 24-bit sample times a float factor would lose the last bit) — restated so that 17 .. 24-bit images come out bit for bit (round 6)
JXL_DEV float int_sample_to_unit(int32_t v, int bits) {
  if (bits < 23) return (float)v * (1.0f / (float)(((uint64_t)1 << bits) - 1));
  return (float)((double)v * (1.0 / (double)(((uint64_t)1 << bits) - 1)));
}
JXL_DEV float sample_bits_to_float(int32_t v, int bits, int exp_bits) {
  uint32_t f = (uint32_t)v;
  if (bits != 32) {
    const int sign_shift = bits - 1, mant_bits = bits - exp_bits - 1, mant_shift = 23 - mant_bits, bias = (1 << (exp_bits - 1)) - 1;
    const uint32_t sign = (f >> sign_shift) & 1u;
    f &= (1u << sign_shift) - 1u;
    if (f == 0) f = sign << 31;
    else {
      int e = (int)(f >> mant_bits);
      uint32_t m = (f & ((1u << mant_bits) - 1u)) << mant_shift;
      if (e == 0 && exp_bits < 8) { while ((m & 0x800000u) == 0) { m <<= 1; e--; } e++; m &= 0x7fffffu; }
      e = e - bias + 127;
      f = (sign << 31) | ((uint32_t)e << 23) | m;
    }
  }
#ifdef __HIPCC__
  return __uint_as_float(f);
#else
  float r; __builtin_memcpy(&r, &f, 4); return r;
#endif
}

// the alpha sample as a float: integer alpha / (2^bits - 1) — `mul`: times the reciprocal instead (the two forms round differently; each call site keeps the
// one its stage of the reference uses) — or the float it encodes
JXL_DEV float alpha_sample_value(const DevFrame &F, int32_t v, bool mul) {
  if (F.mod_alpha_exp_bits) return sample_bits_to_float(v, F.mod_alpha_bits, F.mod_alpha_exp_bits);
  const float mx = (float)((1u << F.mod_alpha_bits) - 1);
  return mul ? (float)v * (1.0f / mx) : (float)v / mx;
}

// ------------------------------------------------------------------ bit reader (LSB first)
// Loads / stores whose address is KNOWN to be HBM (codestream words, frame tables, sample planes), spelled as global-memory accesses: through a generic
// pointer the compiler emits FLAT instructions, which count on the LDS counter too — every later wait for an LDS read then also waits for the memory
// access (hundreds of ns), and a lock-step Modular loop does that several times per sample.  (The CPU harness reads the pointer as it is.)
#ifdef __HIPCC__
template <class T> __device__ __forceinline__ T gld(const T *p) { return *(const T __attribute__((address_space(1))) *)(uintptr_t)p; }
template <class T> __device__ __forceinline__ void gst(T *p, T v) { *(T __attribute__((address_space(1))) *)(uintptr_t)p = v; }
#else
template <class T> inline T gld(const T *p) { return *p; }
template <class T> inline void gst(T *p, T v) { *p = v; }
#endif

struct DevBits {
  const uint32_t *next;     // next aligned word to fetch
  const uint32_t *end;      // first word past the (padded) codestream: reads beyond it yield zeros, never a fault
  uint64_t buf;             // valid bits in the low `n` positions
  uint32_t ahead;           // word fetched one refill early (hides the load latency of the serial lane)
  int32_t n;
  uint64_t consumed;        // bits consumed since the section start
};

JXL_DEV void bits_init(DevBits &b, const uint8_t *base, uint64_t byte_off, uint64_t total_bytes) {
  if (byte_off > total_bytes) byte_off = total_bytes;
  const uint8_t *p = base + byte_off;
  b.end = (const uint32_t *)(base + ((total_bytes + 48) & ~(uint64_t)3));   // buffers carry >= 64 zero bytes of padding
  uint64_t mis = (uint64_t)(uintptr_t)p & 3;
  b.next = (const uint32_t *)(p - mis);
  b.buf = (uint64_t)gld(b.next) | ((uint64_t)gld(b.next + 1) << 32);
  b.ahead = gld(b.next + 2);
  b.next += 3;
  b.buf >>= 8 * mis;
  b.n = 64 - 8 * (int32_t)mis;
  b.consumed = 0;
}
// start `bit_off` bits into the section at byte_off (a stream that follows another one inside the same section)
JXL_DEV void bits_init_at_bit(DevBits &b, const uint8_t *base, uint64_t byte_off, uint64_t bit_off, uint64_t total_bytes) {
  bits_init(b, base, byte_off + (bit_off >> 3), total_bytes);
  const int rem = (int)(bit_off & 7);
  if (rem) { b.buf >>= rem; b.n -= rem; }
  b.consumed = bit_off;
}
JXL_DEV void bits_refill(DevBits &b) {     // guarantees >= 32 valid bits
  if (b.n <= 32) {
    b.buf |= (uint64_t)b.ahead << b.n;
    b.n += 32;
    // (an UNCONDITIONAL load through a clamped pointer — the last word of the buffer's zero padding stands for everything beyond it: with the load under its own
    // condition the compiler materialised the 0, loaded into a scratch register and had to WAIT for the word right here to copy it into `ahead` — the round
    // trip this prefetch exists to hide; measured in the block-form Modular loop, ISA in tools/experiments/block_tree_timers)
    const uint32_t *p = b.next < b.end ? b.next : b.end - 1;
    b.ahead = gld(p);
    b.next++;
  }
}
JXL_DEV uint32_t bits_peek(DevBits &b, int n) {   // n <= 32
  bits_refill(b);
  return (uint32_t)(b.buf & ((1ull << n) - 1));
}
JXL_DEV void bits_skip(DevBits &b, int n) { b.buf >>= n; b.n -= n; b.consumed += (uint64_t)n; }
JXL_DEV uint32_t bits_read(DevBits &b, int n) {   // n <= 32
  if (n == 0) return 0;
  uint32_t v = bits_peek(b, n);
  bits_skip(b, n);
  return v;
}
JXL_DEV uint32_t bits_u32(DevBits &b, int b0, uint32_t o0, int b1, uint32_t o1, int b2, uint32_t o2, int b3, uint32_t o3) {
  uint32_t sel = bits_read(b, 2);
  int nb = sel == 0 ? b0 : sel == 1 ? b1 : sel == 2 ? b2 : b3;
  uint32_t o = sel == 0 ? o0 : sel == 1 ? o1 : sel == 2 ? o2 : o3;
  return nb < 0 ? o : bits_read(b, nb) + o;
}
JXL_DEV int32_t unpack_signed(uint32_t u) { return (int32_t)((u >> 1) ^ (0u - (u & 1))); }

// ------------------------------------------------------------------ symbol reader
struct DevECView {            // resolved pointers for one entropy code
  const uint8_t *ctx_map;
  const uint32_t *cfg;
  const DevAlias *alias;
  const DevPrefix *prefix;
  const uint16_t *pool;
  int32_t use_prefix, log_alpha;
  int32_t lz77, lz_min_symbol, lz_min_length, dist_ctx; uint32_t lz_len_cfg;     // LZ77 (serial walker only)
};

JXL_DEV DevECView ec_view(const uint8_t *tables, const DevEC &e) {
  DevECView v;
  v.ctx_map = tables + e.ctx_map_off;
  v.cfg = (const uint32_t *)(tables + e.cfg_off);
  v.alias = (const DevAlias *)(tables + e.alias_off);
  v.prefix = (const DevPrefix *)(tables + e.prefix_off);
  v.pool = (const uint16_t *)(tables + e.pool_off);
  v.use_prefix = e.use_prefix;
  v.log_alpha = e.log_alpha;
  v.lz77 = e.lz77; v.lz_min_symbol = e.lz_min_symbol; v.lz_min_length = e.lz_min_length; v.dist_ctx = e.num_ctx; v.lz_len_cfg = e.lz_len_cfg;
  return v;
}

JXL_DEV uint32_t ans_init(const DevECView &v, DevBits &b) { return v.use_prefix ? 0x130000u : bits_read(b, 32); }

JXL_DEV uint32_t ec_token(const DevECView &v, DevBits &b, uint32_t &state, uint32_t cluster) {
  if (v.use_prefix) {
    const DevPrefix &p = v.prefix[cluster];
    if (p.single >= 0) return (uint32_t)p.single;
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
      code |= (int)bits_read(b, 1);
      int count = p.cnt[len];
      if (code - first < count) return v.pool[p.sorted_off + (uint32_t)(index + code - first)];
      index += count; first += count; first <<= 1; code <<= 1;
    }
    return 0;
  }
  const int lb = 12 - v.log_alpha;
  uint32_t res = state & 0xfff;
  uint32_t i = res >> lb, pos = res & ((1u << lb) - 1);
  DevAlias e = v.alias[(cluster << v.log_alpha) + i];
  bool right = pos >= e.cutoff;
  uint32_t sym = right ? e.right : i;
  uint32_t off = right ? (uint32_t)e.off1 + pos : pos;
  uint32_t freq = right ? e.freq1 : e.freq0;
  state = freq * (state >> 12) + off;
  if (state < (1u << 16)) state = (state << 16) | bits_read(b, 16);
  return sym;
}

JXL_DEV uint32_t ec_hybrid(DevBits &b, uint32_t cfg, uint32_t token) {
  uint32_t split_exp = cfg & 0xff, msb = (cfg >> 8) & 0xff, lsb = (cfg >> 16) & 0xff;
  uint32_t split = 1u << split_exp;
  if (token < split) return token;
  uint32_t nbits = split_exp - (msb + lsb) + ((token - split) >> (msb + lsb));
  if (nbits > 31) nbits = 31;   // corrupt stream; the final-state check flags it
  uint32_t low = token & ((1u << lsb) - 1);
  token >>= lsb;
  uint32_t bits = bits_read(b, (int)nbits);
  return (((((1u << msb) | (token & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
}

JXL_DEV uint32_t ec_read(const DevECView &v, DevBits &b, uint32_t &state, uint32_t ctx) {
  uint32_t cluster = v.ctx_map[ctx];
  uint32_t token = ec_token(v, b, state, cluster);
  return ec_hybrid(b, v.cfg[cluster], token);
}


// ---- LZ77 (C.3.3).  Used by the serial Modular walker for codes that enable it (libjxl's low-effort lossless encoders); the window
// holds the decoded integers of the stream, min(2^20, symbols of the stream) entries in HBM.
JXL_CONST int8_t kLzSpecialDist[120][2] = {
    {0, 1},  {1, 0},  {1, 1},  {-1, 1}, {0, 2},  {2, 0},  {1, 2},  {-1, 2}, {2, 1},  {-2, 1}, {2, 2},  {-2, 2},
    {0, 3},  {3, 0},  {1, 3},  {-1, 3}, {3, 1},  {-3, 1}, {2, 3},  {-2, 3}, {3, 2},  {-3, 2}, {0, 4},  {4, 0},
    {1, 4},  {-1, 4}, {4, 1},  {-4, 1}, {3, 3},  {-3, 3}, {2, 4},  {-2, 4}, {4, 2},  {-4, 2}, {0, 5},  {3, 4},
    {-3, 4}, {4, 3},  {-4, 3}, {5, 0},  {1, 5},  {-1, 5}, {5, 1},  {-5, 1}, {2, 5},  {-2, 5}, {5, 2},  {-5, 2},
    {4, 4},  {-4, 4}, {3, 5},  {-3, 5}, {5, 3},  {-5, 3}, {0, 6},  {6, 0},  {1, 6},  {-1, 6}, {6, 1},  {-6, 1},
    {2, 6},  {-2, 6}, {6, 2},  {-6, 2}, {4, 5},  {-4, 5}, {5, 4},  {-5, 4}, {3, 6},  {-3, 6}, {6, 3},  {-6, 3},
    {0, 7},  {7, 0},  {1, 7},  {-1, 7}, {5, 5},  {-5, 5}, {7, 1},  {-7, 1}, {4, 6},  {-4, 6}, {6, 4},  {-6, 4},
    {2, 7},  {-2, 7}, {7, 2},  {-7, 2}, {3, 7},  {-3, 7}, {7, 3},  {-7, 3}, {5, 6},  {-5, 6}, {6, 5},  {-6, 5},
    {8, 0},  {4, 7},  {-4, 7}, {7, 4},  {-7, 4}, {8, 1},  {8, 2},  {6, 6},  {-6, 6}, {8, 3},  {5, 7},  {-5, 7},
    {7, 5},  {-7, 5}, {8, 4},  {6, 7},  {-6, 7}, {7, 6},  {-7, 6}, {8, 5},  {7, 7},  {-7, 7}, {8, 6},  {8, 7}};
struct DevLz { uint32_t *win; uint32_t win_len, ncopy, pos, ndec, dist_mult, err; };
JXL_DEV uint32_t ec_read_lz(const DevECView &v, DevBits &b, uint32_t &state, uint32_t ctx, DevLz &z) {
  const uint32_t mask = (1u << 20) - 1;
  for (int guard = 0; guard < 2; guard++) {
    if (z.ncopy) {
      const uint32_t r = z.win[(z.pos++) & mask];
      z.ncopy--;
      z.win[(z.ndec++) & mask] = r;
      return r;
    }
    const uint32_t cluster = v.ctx_map[ctx];
    const uint32_t token = ec_token(v, b, state, cluster);
    if (token < (uint32_t)v.lz_min_symbol) {
      const uint32_t r = ec_hybrid(b, v.cfg[cluster], token);
      if (z.win_len == (1u << 20) || z.ndec < z.win_len) z.win[z.ndec & mask] = r;      // a full-size window wraps; a shorter one holds the whole stream
      z.ndec++;
      return r;
    }
    uint32_t ncopy = ec_hybrid(b, v.lz_len_cfg, token - (uint32_t)v.lz_min_symbol) + (uint32_t)v.lz_min_length;
    const uint32_t dc = v.ctx_map[v.dist_ctx];
    const uint32_t dtok = ec_token(v, b, state, dc);
    uint32_t distance = ec_hybrid(b, v.cfg[dc], dtok);
    const uint32_t nspecial = z.dist_mult ? 120u : 0u;
    if (distance < nspecial) {
      const int d = (int)z.dist_mult * kLzSpecialDist[distance][1] + kLzSpecialDist[distance][0];
      distance = d < 1 ? 1u : (uint32_t)d;
    } else distance = distance + 1 - nspecial;
    if (distance > z.ndec) distance = z.ndec;
    if (distance > (1u << 20)) distance = 1u << 20;
    z.pos = z.ndec - distance;
    if (distance == 0) {                      // nothing decoded yet: the copy source reads as zeros
      uint32_t n = ncopy < z.win_len ? ncopy : z.win_len;
      for (uint32_t i = 0; i < n; i++) z.win[i] = 0;
    }
    // a copy can never run past the symbols the stream holds (the window has exactly that many entries when shorter than 2^20)
    if (ncopy < (uint32_t)v.lz_min_length || (z.win_len < (1u << 20) && (uint64_t)z.ndec + ncopy > z.win_len)) { z.err = 1; return 0; }
    z.ncopy = ncopy;
  }
  z.err = 1;
  return 0;
}

}  // namespace jxlamd
