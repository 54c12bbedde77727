/* TEST INFRASTRUCTURE (tests/): the complete est-fact host program with the CPU ORACLE as the
 * compute backend, so that the host logic can be compared with the reference in a container that
 * has no GPU.  Never shipped: the product est-fact links the GPU backend only. */
#include <stdlib.h>
#include <time.h>
#include <string.h>

#include "../../pintron_amd/host/estfact.h"
#include "../../oracle/dp_oracle.h"
#include "../../oracle/pairing_oracle.h"

/* ESTFACT_CHECK_TIMING=1: seconds spent inside the oracle backend vs. the whole run, to stderr */
static double backend_s;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int oracle_pairings(void* self, const char* pattern, size_t m, unsigned L, double rate,
                           ef_triple** out, size_t* n) {
  const double t0 = now_s();
  long cap = 4096;
  int32_t* buf = (int32_t*)malloc(3 * cap * sizeof(int32_t));
  long cnt = orc_pairings((const orc_index*)self, pattern, m, L, rate, buf, cap, NULL);
  if (cnt > cap) { cap = cnt; buf = (int32_t*)realloc(buf, 3 * cap * sizeof(int32_t)); cnt = orc_pairings((const orc_index*)self, pattern, m, L, rate, buf, cap, NULL); }
  *out = (ef_triple*)buf; *n = (size_t)cnt;
  backend_s += now_s() - t0;
  return 0;
}

/* ESTFACT_CHECK_DIMS=1: the largest operands per DP kind, to stderr (what the device limits of
 * include/pintron_gpu.h have to cover on real inputs) */
static size_t dim_max[8][2];
static int oracle_dp_impl(void* self, const ef_dp_req* q, ef_dp_res* r);
static int oracle_dp(void* self, const ef_dp_req* q, ef_dp_res* r) {
  if ((unsigned)q->kind < 8) {
    if (q->la > dim_max[q->kind][0]) dim_max[q->kind][0] = q->la;
    if (q->lb > dim_max[q->kind][1]) dim_max[q->kind][1] = q->lb;
  }
  const double t0 = now_s();
  const int rc = oracle_dp_impl(self, q, r);
  backend_s += now_s() - t0;
  return rc;
}
static int oracle_dp_impl(void* self, const ef_dp_req* q, ef_dp_res* r) {
  (void)self;
  switch (q->kind) {
    case EF_DP_ALIGN: {
      ef_dp_res_rows(r, q->la + q->lb + 2);
      int32_t dim;
      r->v[0] = (int32_t)orc_align(q->a, q->la, q->b, q->lb, r->s0, r->s1, &dim);
      r->v[1] = dim;
      return 0;
    }
    case EF_DP_GAP: {
      ef_dp_res_rows(r, q->la + q->lb + 2);
      orc_gap_result g;
      orc_gap_align(q->a, q->la, q->b, q->lb, r->s0, r->s1, &g);
      r->v[0] = g.dim; r->v[1] = g.factor_cut; r->v[2] = g.intron_start; r->v[3] = g.intron_end;
      r->v[4] = g.intron_start_on_align; r->v[5] = g.intron_end_on_align;
      return 0;
    }
    case EF_DP_ED: r->v[0] = (int32_t)orc_edit_distance(q->a, q->la, q->b, q->lb); return 0;
    case EF_DP_KBAND: {
      uint32_t e; r->v[0] = orc_kband(q->a, q->la, q->b, q->lb, q->p0, &e); r->v[1] = (int32_t)e;
      if (q->tail) { double thr; const uint32_t w[2] = { q->p1, q->p2 }; memcpy(&thr, w, 8); r->v[2] = (int32_t)orc_dust_flags(q->a, q->la, q->b, q->lb, thr); }
      return 0;
    }
    case EF_DP_LCF: { uint32_t o1, o2, ln; orc_lcf(q->a, q->la, q->b, q->lb, &o1, &o2, &ln); r->v[0] = (int32_t)ln; r->v[1] = (int32_t)o1; r->v[2] = (int32_t)o2; return 0; }
    case EF_DP_BORDERS: {
      char* t = (char*)calloc(q->lb + 3, 1);
      memcpy(t, q->b, q->lb + (q->tail > 2 ? 2 : q->tail));
      orc_borders_result b;
      orc_refine_borders(q->a, q->la, q->p0, q->p1, t, q->lb, q->p2, &b);
      free(t);
      r->v[0] = b.ok; r->v[1] = (int32_t)b.offset_p; r->v[2] = (int32_t)b.offset_t1; r->v[3] = (int32_t)b.offset_t2; r->v[4] = (int32_t)b.edit_distance;
      return 0;
    }
    case EF_DP_AFFIX: { uint32_t e = 0, g = 0; r->v[0] = orc_longest_affix(q->a, q->la, q->b, q->lb, &e, &g); r->v[1] = (int32_t)e; r->v[2] = (int32_t)g; return 0; }
  }
  return -1;
}

static ef_backend* open_oracle(const ef_seq* gen) {
  ef_backend* be = (ef_backend*)calloc(1, sizeof(ef_backend));
  const double t0 = now_s();
  be->self = orc_index_create(gen->seq, strlen(gen->seq));
  backend_s += now_s() - t0;
  be->pairings = oracle_pairings;
  be->dp = oracle_dp;
  return be;
}
static void close_oracle(ef_backend* be) { orc_index_destroy((orc_index*)be->self); free(be); }

int main(int argc, char** argv) {
  const double t0 = now_s();
  const int rc = ef_run(argc, argv, open_oracle, close_oracle);
  if (getenv("ESTFACT_CHECK_DIMS")) fprintf(stderr, "work: high water %llu units\n", ef_work_high_water());
  if (getenv("ESTFACT_CHECK_DIMS"))
    for (int k = 0; k < 8; ++k) if (dim_max[k][0] || dim_max[k][1]) fprintf(stderr, "dims: kind %d max a_len %zu max b_len %zu\n", k, dim_max[k][0], dim_max[k][1]);
  if (getenv("ESTFACT_CHECK_TIMING")) fprintf(stderr, "timing: total %.3f s, oracle backend %.3f s, host logic %.3f s\n", now_s() - t0, backend_s, now_s() - t0 - backend_s);
  return rc;
}
