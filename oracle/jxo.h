/* oracle/jxo.h — CPU restatement (plain scalar C) of the JPEG XL decode path that the reference
 * runs inside libjxl's JxlDecoderProcessInput (reference call site:
 * jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75; public C API jxlcoder/src/main/cpp/jxl/decode.h).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (jxl_coder_amd/) may include, link or call this.
 * libjxl's source is NOT part of the reference tree (build_jxl.sh:6-11 clones github.com/libjxl/libjxl
 * at HEAD; the vendored binary reports 0.12.0), so this file restates the published algorithm
 * (ISO/IEC 18181-1) and is pinned against the reference's own prebuilt libjxl.so through
 * oracle/_ref (see oracle/README.md): every stage below was validated on pixels, not on source.
 */
#ifndef JXO_H_
#define JXO_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- bit reader */
typedef struct {
  const uint8_t *p;
  size_t len;    /* bytes */
  size_t pos;    /* bit position */
  int err;       /* sticky: read past end */
} jxo_br;

void jxo_br_init(jxo_br *br, const uint8_t *p, size_t len);
uint32_t jxo_bits(jxo_br *br, int n);           /* n in 0..32, LSB first */
static inline int jxo_bool(jxo_br *br) { return (int)jxo_bits(br, 1); }
void jxo_align(jxo_br *br);                      /* ZeroPadToByte */
/* U32(d0,d1,d2,d3): each distribution is (nbits, offset); nbits<0 means Val(offset) */
uint32_t jxo_u32(jxo_br *br, int b0, uint32_t o0, int b1, uint32_t o1, int b2, uint32_t o2, int b3, uint32_t o3);
uint64_t jxo_u64(jxo_br *br);
float jxo_f16(jxo_br *br);
uint32_t jxo_enum(jxo_br *br);
static inline int32_t jxo_unpack_signed(uint32_t u) { return (int32_t)((u >> 1) ^ (0u - (u & 1))); }

/* ---------------------------------------------------------------- entropy code (ANS / prefix, hybrid uint, LZ77) */
typedef struct { uint8_t split_exp, msb, lsb; } jxo_huc;

typedef struct {
  uint16_t cutoff, offsets1;   /* offsets1 may exceed 16 bit? no: < 4096*? keep 32 below */
} jxo_alias_unused;

typedef struct {
  /* ANS */
  uint16_t *D;          /* [1<<log_alpha] frequencies (12-bit total 4096) */
  uint8_t *a_sym;       /* alias: right_value */
  uint16_t *a_cutoff;
  uint32_t *a_off;      /* offsets1 */
  /* prefix */
  uint16_t cnt[16];     /* number of codes of each length */
  uint16_t *sorted;     /* symbols sorted by (len, value) */
  int nsym;             /* alphabet size for prefix code */
  int single;           /* prefix: only symbol (if alphabet or code degenerate) else -1 */
} jxo_cluster;

typedef struct {
  int num_ctx;          /* number of contexts (before the lz77 extra one) */
  uint8_t *ctx_map;     /* [num_ctx (+1 if lz77)] */
  int num_clusters;
  int use_prefix;
  int log_alpha;
  jxo_huc *cfg;         /* per cluster */
  jxo_cluster *cl;
  int lz77, lz_min_symbol, lz_min_length;
  jxo_huc lz_len_cfg;
  /* reader state */
  uint32_t state;
  uint32_t *window;     /* lz77 window, 1<<20 */
  uint32_t num_to_copy, copy_pos, num_decoded;
  uint32_t dist_mult;
} jxo_ec;

int jxo_ec_read_header(jxo_ec *ec, jxo_br *br, int num_ctx);     /* 0 ok */
void jxo_ec_begin(jxo_ec *ec, jxo_br *br, uint32_t dist_mult);   /* reads ANS state */
uint32_t jxo_ec_read(jxo_ec *ec, jxo_br *br, int ctx);           /* hybrid uint */
int jxo_ec_final_ok(const jxo_ec *ec);                           /* ANS final state == 0x130000 */
void jxo_ec_free(jxo_ec *ec);
/* Lehmer-coded permutation (coefficient orders, TOC). out[size]. returns 0 ok */
int jxo_read_permutation(jxo_ec *ec, jxo_br *br, uint32_t *out, uint32_t size, uint32_t skip);

/* ---------------------------------------------------------------- public result */
typedef struct {
  uint32_t xsize, ysize;          /* oriented output dims */
  uint32_t bits_per_sample, exp_bits;
  uint32_t num_color_channels, num_extra_channels, alpha_bits, alpha_premultiplied;
  uint32_t orientation;           /* as libjxl reports after re-orientation: 1 */
  uint32_t have_animation, xyb_encoded;
  float intensity_target;
  uint32_t want_icc;
  uint32_t color_space, white_point, primaries, transfer_function, rendering_intent;
  uint32_t have_gamma; float gamma;
} jxo_info;

/* Decode to interleaved RGBA, 8 or 16 bit per sample (out_bits), as the reference driver asks of libjxl.
 * returns 0 on success, <0 on error/unsupported (message in jxo_last_error()). Caller frees *out with free(). */
int jxo_decode(const uint8_t *data, size_t size, int out_bits, uint8_t **out, size_t *out_size, jxo_info *info);
int jxo_basic_info(const uint8_t *data, size_t size, jxo_info *info);
const char *jxo_last_error(void);

/* debug: dump intermediate float planes (XYB after IDCT etc.) */
extern int jxo_debug;

#ifdef __cplusplus
}
#endif
#endif
