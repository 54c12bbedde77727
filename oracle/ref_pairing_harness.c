/*
 * TEST INFRASTRUCTURE ONLY (oracle/): calls the reference's own index construction and
 * build_vertex_set (src/max-emb-graph.c:218) from libpintron_ref_core.so and flattens the vertex set,
 * so the pairing oracle and the HIP pairing kernel can be compared with the reference per EST.
 * The call sequence is the one of src/main-est-fact.c:224-239 and src/compute-est-fact.c:101-107.
 */
#include <stdlib.h>
#include <string.h>

#include "types.h"
#include "list.h"
#include "ext_array.h"
#include "util.h"
#include "configuration.h"
#include "aug_suffix_tree.h"
#include "max-emb-graph.h"

/* build_vertex_set and stree_preprocess take a `pconfiguration`; the harness owns a plain
 * struct _configuration (include/configuration.h:39-135) and sets the fields they read:
 * min_factor_len and min_string_depth_rate (src/max-emb-graph.c:273-274, src/aug_suffix_tree.c:247). */
static pconfiguration harness_config(void) {
  pconfiguration c = (pconfiguration)calloc(1, sizeof(struct _configuration));
  c->min_factor_len = 15;            /* src/options.ggo:94-101 */
  c->min_string_depth_rate = 0.2;    /* :124-137 */
  return c;
}

typedef struct {
  pEST_info gen;
  LST_STree* tree;
  ppreproc_gen pg;
  pconfiguration cfg;
} ref_index;

void* ref_index_create(const char* genomic) {
  ref_index* ix = (ref_index*)calloc(1, sizeof(ref_index));
  ix->gen = EST_info_create();
  ix->gen->EST_seq = alloc_and_copy(genomic);
  ix->gen->EST_id = alloc_and_copy(">harness");
  ix->cfg = harness_config();
  LST_StringSet* set = lst_stringset_new();
  LST_String* lst = PALLOC(LST_String);
  lst_string_init(lst, ix->gen->EST_seq, sizeof(char), strlen(ix->gen->EST_seq));
  lst_stringset_add(set, lst);
  ix->tree = lst_stree_new(set);
  ix->pg = PGen_create();
  preprocess_text(ix->gen, ix->pg);
  stree_preprocess(ix->tree, ix->pg, ix->cfg);
  return ix;
}

/* triples (p,t,l) of every position list, in list order, source and sink excluded.
 * Returns the number of pairings (may exceed cap; only cap are written). */
long ref_build_pairings(void* index, const char* est_seq, unsigned min_factor_len,
                        double min_string_depth_rate, int* out, long cap) {
  ref_index* ix = (ref_index*)index;
  pEST_info est = EST_info_create();
  est->EST_seq = alloc_and_copy(est_seq);
  est->EST_id = alloc_and_copy(">est");
  struct _configuration cfg = *ix->cfg;
  cfg.min_factor_len = min_factor_len;
  cfg.min_string_depth_rate = min_string_depth_rate;
  pext_array V = build_vertex_set(est, ix->tree, ix->pg, &cfg);
  long n = 0;
  const size_t sz = EA_size(V);
  for (size_t i = 1; i + 1 < sz; ++i) {
    plist Vi = (plist)EA_get(V, (int)i);
    plistit it = list_first(Vi);
    while (listit_has_next(it)) {
      ppairing p = (ppairing)listit_next(it);
      if (n < cap) { out[3 * n] = p->p; out[3 * n + 1] = p->t; out[3 * n + 2] = p->l; }
      ++n;
    }
    listit_destroy(it);
  }
  return n;
}
