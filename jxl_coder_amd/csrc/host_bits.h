// jxl_coder_amd/csrc/host_bits.h — host-side bit reader and entropy-code HEADER parser (ISO/IEC 18181-1
// Annex C/D).  The host only decodes what is global and tiny (TOC permutation, context maps, MA tree,
// coefficient orders); the per-group streams are decoded on the GPU from the tables packed here.
// Part of the product's host code (C++), replaces what libjxl does before its group loops
// (reference call site jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).
#ifndef HX_H_
#define HX_H_
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
} hx_br;

void hx_br_init(hx_br *br, const uint8_t *p, size_t len);
uint32_t hx_bits(hx_br *br, int n);           /* n in 0..32, LSB first */
static inline int hx_bool(hx_br *br) { return (int)hx_bits(br, 1); }
void hx_align(hx_br *br);                      /* ZeroPadToByte */
/* U32(d0,d1,d2,d3): each distribution is (nbits, offset); nbits<0 means Val(offset) */
uint32_t hx_u32(hx_br *br, int b0, uint32_t o0, int b1, uint32_t o1, int b2, uint32_t o2, int b3, uint32_t o3);
uint64_t hx_u64(hx_br *br);
float hx_f16(hx_br *br);
uint32_t hx_enum(hx_br *br);
static inline int32_t hx_unpack_signed(uint32_t u) { return (int32_t)((u >> 1) ^ (0u - (u & 1))); }

/* ---------------------------------------------------------------- entropy code (ANS / prefix, hybrid uint, LZ77) */
typedef struct { uint8_t split_exp, msb, lsb; } hx_huc;

typedef struct {
  uint16_t cutoff, offsets1;   /* offsets1 may exceed 16 bit? no: < 4096*? keep 32 below */
} hx_alias_unused;

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
} hx_cluster;

typedef struct {
  int num_ctx;          /* number of contexts (before the lz77 extra one) */
  uint8_t *ctx_map;     /* [num_ctx (+1 if lz77)] */
  int num_clusters;
  int use_prefix;
  int log_alpha;
  hx_huc *cfg;         /* per cluster */
  hx_cluster *cl;
  int lz77, lz_min_symbol, lz_min_length;
  hx_huc lz_len_cfg;
  /* reader state */
  uint32_t state;
  uint32_t *window;     /* lz77 window, 1<<20 */
  uint32_t num_to_copy, copy_pos, num_decoded;
  uint32_t dist_mult;
} hx_ec;

int hx_ec_read_header(hx_ec *ec, hx_br *br, int num_ctx);     /* 0 ok */
void hx_ec_begin(hx_ec *ec, hx_br *br, uint32_t dist_mult);   /* reads ANS state */
uint32_t hx_ec_read(hx_ec *ec, hx_br *br, int ctx);           /* hybrid uint */
int hx_ec_final_ok(const hx_ec *ec);                           /* ANS final state == 0x130000 */
void hx_ec_free(hx_ec *ec);
/* Lehmer-coded permutation (coefficient orders, TOC). out[size]. returns 0 ok */
int hx_read_permutation(hx_ec *ec, hx_br *br, uint32_t *out, uint32_t size, uint32_t skip);


void hx_set_error(const char *fmt, ...);
const char *hx_last_error(void);
#define HX_FAIL(...) do { hx_set_error(__VA_ARGS__); return -1; } while (0)
int hx__read_ctx_map(hx_br *br, uint8_t *map, int n, int *num_clusters);
#ifdef __cplusplus
}
#endif
#endif
