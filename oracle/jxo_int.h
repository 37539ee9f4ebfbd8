/* oracle/jxo_int.h — internal structures of the CPU restatement (test infrastructure only, see jxo.h). */
#ifndef JXO_INT_H_
#define JXO_INT_H_
#include "jxo.h"

typedef struct {
  int w, h, hshift, vshift;
  int32_t *d;                 /* row-major w*h */
} jxo_chan;

typedef struct {
  int prop;                   /* -1 = leaf */
  int32_t splitval;
  int lchild, rchild;
  int predictor;
  int64_t offset;
  uint32_t multiplier;
  int ctx;                    /* leaf id */
} jxo_tnode;

typedef struct {
  jxo_tnode *n;
  int count;
  int num_leaves;
  jxo_ec code;
  int valid;
} jxo_tree;

typedef struct {
  int p1, p2, p3a, p3b, p3c, p3d, p3e;
  int w[4];
} jxo_wp_header;

enum { JXO_TR_RCT = 0, JXO_TR_PALETTE = 1, JXO_TR_SQUEEZE = 2 };
typedef struct { unsigned char horizontal, in_place; int begin_c, num_c; } jxo_squeeze_step;
typedef struct {
  int id;
  int begin_c, rct_type;
  int num_c, nb_colours, nb_deltas, d_pred;
  int nsq; jxo_squeeze_step sq[48];      /* squeeze (H.6.2): the explicit steps, or the default sequence resolved at meta-apply time */
} jxo_transform;

typedef struct {
  jxo_chan *ch;
  int nch, cap;
  int nb_meta;
  int bitdepth;
  jxo_transform tr[64];
  int ntr;
} jxo_modimg;

void jxo_modimg_init(jxo_modimg *m);
int jxo_modimg_add(jxo_modimg *m, int w, int h, int hshift, int vshift);   /* appends, returns index */
void jxo_modimg_free(jxo_modimg *m);

int jxo_tree_read(jxo_tree *t, jxo_br *br);
void jxo_tree_free(jxo_tree *t);

/* Decode one modular sub-bitstream into img's channels (all channels that are not "too large").
 * max_chan_size: channels (after meta) with w or h above it stop the loop (GlobalModular); 0 = no limit.
 * undo_transforms: apply inverse transforms at the end. returns 0 ok. */
int jxo_modular_decode(jxo_br *br, jxo_modimg *img, int stream_id, int max_chan_size, jxo_tree *global_tree,
                       int undo_transforms, int *first_undecoded);
/* header part only is read inside; for full-frame modular the caller needs the transform list: kept in img->tr */
int jxo_modular_undo_transforms(jxo_modimg *img);

void jxo_set_error(const char *fmt, ...);
#define JXO_FAIL(...) do { jxo_set_error(__VA_ARGS__); return -1; } while (0)

#endif
