/* TEST INFRASTRUCTURE (tests/): runs the host-side MEG construction of the est-fact program on
 * genomic.txt / ests.txt of the current directory with the CPU pairing ORACLE as the backend, and
 * writes megs-check.txt in the reference's megs.txt record format for EVERY entry of the EST
 * list (both strands).  With MEG_CHECK_FIRST_ATTEMPT=1 it writes megs-first.txt instead: for every
 * entry the graph of the FIRST attempt of build_meg (no retry with a longer factor), as
 *   "@@seq <prepared sequence>\n@@complex <0|1>\n<meg_write text>@@edges\n<meg-edges text>@@end\n"
 * which is what the device's MEG stage returns per pattern (tests/test_gpu_pairings.py).
 * The product binary never links this file nor the oracle. */
#include <stdlib.h>
#include <string.h>

#include "../../pintron_amd/host/estfact.h"
#include "../../oracle/pairing_oracle.h"

static int oracle_pairings(void* self, const char* pattern, size_t m, unsigned L, double rate,
                           ef_triple** out, size_t* n) {
  long cap = 4096;
  int32_t* buf = (int32_t*)malloc(3 * cap * sizeof(int32_t));
  long cnt = orc_pairings((const orc_index*)self, pattern, m, L, rate, buf, cap, NULL);
  if (cnt > cap) {
    cap = cnt; buf = (int32_t*)realloc(buf, 3 * cap * sizeof(int32_t));
    cnt = orc_pairings((const orc_index*)self, pattern, m, L, rate, buf, cap, NULL);
  }
  *out = (ef_triple*)buf; *n = (size_t)cnt;
  return 0;
}

int main(int argc, char** argv) {
  ef_config cfg;
  if (ef_config_load(&cfg, argc, argv) != 0) return 2;
  ef_seq** gens; ef_seq** ests;
  if (ef_read_multifasta("genomic.txt", &gens) != 1) { fprintf(stderr, "genomic.txt: need exactly one sequence\n"); return 1; }
  ef_seq* gen = gens[0];
  ef_parse_genomic_header(gen);
  if (ef_ntails_removal(gen) != 0) return 1;
  const long n = ef_read_multifasta("ests.txt", &ests);
  if (n < 0) return 1;
  orc_index* ix = orc_index_create(gen->seq, strlen(gen->seq));
  if (getenv("MEG_CHECK_FIRST_ATTEMPT")) {         /* the genomic sequence as the index sees it */
    FILE* gp = fopen("genomic-prepared.txt", "w");
    fputs(gen->seq, gp);
    fclose(gp);
  }
  ef_backend be;
  memset(&be, 0, sizeof be);
  be.self = ix; be.pairings = oracle_pairings;
  ef_sink sink = { fopen("megs-check.txt", "w"), NULL, 0, 0 };
  ef_sink* f = &sink;
  for (long i = 0; i < n; ++i) {
    ef_seq* est = ests[i];
    ef_set_gb_identification(est);
    ef_set_strand_and_rc(est);
    ef_polyAT_substitution(est);
    ef_seq* both[2] = { est, NULL };
    if (!est->fixed_strand) { both[1] = ef_copy_and_reverse(est); ef_polyAT_substitution(both[1]); }
    for (int k = 0; k < 2 && both[k]; ++k) {
      if (getenv("MEG_CHECK_FIRST_ATTEMPT")) {
        static ef_sink first;
        if (!first.f) first.f = fopen("megs-first.txt", "w");
        ef_triple* tr = NULL; size_t ntr = 0;
        const size_t m = strlen(both[k]->seq);
        oracle_pairings(ix, both[k]->seq, m, cfg.min_factor_len, cfg.min_string_depth_rate, &tr, &ntr);
        ef_meg* V = ef_meg_from_pairings(tr, ntr, m);
        free(tr);
        ef_build_edge_set(V, &cfg);
        ef_simplify_meg(V, &cfg);
        if (cfg.trans_red) ef_transitive_reduction(V);
        bool cx = ef_is_too_complex_for_compaction(V);
        if (!cx && cfg.short_edge_comp) ef_compact_short_edges(V, &cfg);
        cx = cx || ef_is_too_complex(V, &cfg);
        ef_sink_puts(&first, "@@seq "); ef_sink_puts(&first, both[k]->seq);
        ef_sink_puts(&first, cx ? "\n@@complex 1\n" : "\n@@complex 0\n");
        ef_meg_write(&first, V);
        ef_sink_puts(&first, "@@edges\n");
        ef_intronic_edges_write(&first, V);
        ef_sink_puts(&first, "@@end\n");
        ef_meg_free(V);
        continue;
      }
      size_t inc = 0;
      ef_meg* V = ef_build_meg(both[k], &be, &cfg, &inc);
      ef_sink_puts(f, "\n\n***********\n\n");
      ef_write_single_est_info(f, both[k]);
      ef_meg_write(f, V);
      ef_meg_free(V);
    }
  }
  fclose(sink.f);
  return 0;
}
