/*
 * oracle/pairing_oracle.h -- TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the reference's MEG vertex-set construction: build_vertex_set
 * (src/max-emb-graph.c:218-392) over the augmented suffix tree (src/aug_suffix_tree.c),
 * re-expressed over a suffix array + LCP array.  Pinned against the reference's own
 * build_vertex_set (oracle/ref_pairing_harness.c) in tests/test_pairings_oracle.py.
 */
#ifndef PINTRON_PAIRING_ORACLE_H
#define PINTRON_PAIRING_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;
orc_index* orc_index_create(const char* genomic, size_t n);
void orc_index_destroy(orc_index* ix);
const uint32_t* orc_index_sa(const orc_index* ix);      /* n entries */
const uint32_t* orc_index_lcp(const orc_index* ix);     /* n+1 entries, lcp[0] = lcp[n] = 0 */

/* (p,t,l) triples of all positions in the reference's final list order.  Returns the number of
 * pairings (only min(count, cap) are written).  depth_out (optional, m entries) receives the
 * emulated locus depth D_i of every position (0 when below min_factor_len). */
long orc_pairings(const orc_index* ix, const char* pattern, size_t m, uint32_t min_factor_len,
                  double min_string_depth_rate, int32_t* out, long cap, int32_t* depth_out);

#ifdef __cplusplus
}
#endif
#endif
