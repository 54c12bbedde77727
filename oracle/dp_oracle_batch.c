/*
 * oracle/dp_oracle_batch.c -- TEST INFRASTRUCTURE ONLY (see dp_oracle.h).
 * Runs a batch of jobs in the C-ABI's own job/result layout (include/pintron_gpu.h is included
 * for the struct definitions only; nothing from the HIP library is linked) through the CPU
 * oracle, single-threaded.  Used by the parity tests to compare whole result arrays and by
 * bench.py's cpu_baseline leg ("port").
 */
#include <string.h>
#include <stdlib.h>

#include "../include/pintron_gpu.h"
#include "dp_oracle.h"

/* returns the number of DP cells evaluated with the reference's loop bounds */
uint64_t orc_dp_batch(const pgpu_dp_job* jobs, size_t n, const char* arena, const char* genomic,
                      pgpu_dp_result* results, char* strings, size_t strings_cap,
                      size_t* strings_used) {
  uint64_t cells = 0;
  size_t soff = 0;
  for (size_t i = 0; i < n; ++i) {
    const pgpu_dp_job* j = &jobs[i];
    pgpu_dp_result* r = &results[i];
    memset(r, 0, sizeof(*r));
    const char* a = ((j->flags & PGPU_JOB_A_GENOMIC) ? genomic : arena) + j->a_off;
    const char* b = ((j->flags & PGPU_JOB_B_GENOMIC) ? genomic : arena) + j->b_off;
    const size_t la = j->a_len, lb = j->b_len;
    switch (j->kind) {
      case PGPU_DP_ALIGN: {
        const size_t cap = la + lb + 1;
        if (soff + 2 * cap > strings_cap) { r->status = PGPU_ENOSPC; break; }
        int32_t dim;
        r->v[0] = (int32_t)orc_align(a, la, b, lb, strings + soff, strings + soff + cap, &dim);
        r->v[1] = dim;
        r->str[0] = soff; r->str[1] = soff + cap;
        soff += 2 * cap;
        cells += orc_cells_align(a, la, b, lb);
        break;
      }
      case PGPU_DP_GAP: {
        const size_t cap = la + lb + 1;
        if (soff + 2 * cap > strings_cap) { r->status = PGPU_ENOSPC; break; }
        orc_gap_result g;
        orc_gap_align(a, la, b, lb, strings + soff, strings + soff + cap, &g);
        r->v[0] = g.dim; r->v[1] = g.factor_cut; r->v[2] = g.intron_start; r->v[3] = g.intron_end;
        r->v[4] = g.intron_start_on_align; r->v[5] = g.intron_end_on_align;
        r->str[0] = soff; r->str[1] = soff + cap;
        soff += 2 * cap;
        cells += 3ull * la * lb;
        break;
      }
      case PGPU_DP_ED:
        r->v[0] = (int32_t)orc_edit_distance(a, la, b, lb);
        cells += (uint64_t)la * lb;
        break;
      case PGPU_DP_KBAND: {
        uint32_t e;
        r->v[0] = orc_kband(a, la, b, lb, j->p0, &e);
        r->v[1] = (int32_t)e;
        if (j->tail) { double thr; const uint32_t w[2] = { j->p1, j->p2 }; memcpy(&thr, w, 8); r->v[2] = (int32_t)orc_dust_flags(a, la, b, lb, thr); }
        cells += orc_cells_kband(a, la, b, lb, j->p0);
        break;
      }
      case PGPU_DP_LCF: {
        uint32_t o1, o2, ln;
        orc_lcf(a, la, b, lb, &o1, &o2, &ln);
        r->v[0] = (int32_t)ln; r->v[1] = (int32_t)o1; r->v[2] = (int32_t)o2;
        cells += (uint64_t)la * lb;
        break;
      }
      case PGPU_DP_BORDERS: {
        /* the oracle reads up to two bytes past t like the reference: give it a private copy
         * with exactly `tail` valid bytes followed by NULs */
        char* t = (char*)malloc(lb + 3);
        const size_t tail = j->tail > 2 ? 2 : j->tail;
        memcpy(t, b, lb + tail);
        t[lb + tail] = '\0'; t[lb + 2] = '\0';
        if (tail < 1) t[lb + 1] = '\0';
        orc_borders_result br;
        orc_refine_borders(a, la, j->p0, j->p1, t, lb, j->p2, &br);
        free(t);
        r->v[0] = br.ok; r->v[1] = (int32_t)br.offset_p; r->v[2] = (int32_t)br.offset_t1;
        r->v[3] = (int32_t)br.offset_t2; r->v[4] = (int32_t)br.edit_distance;
        const size_t tw = la + j->p2 < lb ? la + j->p2 : lb;
        cells += 2ull * la * tw;
        break;
      }
      case PGPU_DP_AFFIX: {
        uint32_t e = 0, g = 0;
        r->v[0] = orc_longest_affix(a, la, b, lb, &e, &g);
        r->v[1] = r->v[0] ? (int32_t)e : 0; r->v[2] = r->v[0] ? (int32_t)g : 0;
        cells += (uint64_t)la * lb;
        break;
      }
      default: r->status = PGPU_EINVAL;
    }
  }
  if (strings_used) *strings_used = soff;
  return cells;
}
