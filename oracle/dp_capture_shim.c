/*
 * TEST INFRASTRUCTURE ONLY (oracle/): LD_PRELOAD interposer for the reference est-fact.
 *
 * Records every call the *unmodified* reference (oracle/_ref/est-fact-core) makes to its exported
 * DP routines -- inputs and outputs -- as JSON lines in $PINTRON_DP_CAPTURE.  Used by
 * tools/make_golden.py to produce tests/golden/dp_calls_*.jsonl and the DP job census.
 * The reference sources are not touched: the calls are intercepted at the dynamic-linker level
 * (the routines are default-visibility symbols of libpintron_ref_core.so reached through the PLT).
 *
 * Interposed: compute_alignment (src/compute-alignments.c:39), compute_gap_alignment
 * (src/refine-intron.c:560), edit_distance (src/refine.c:50), compute_edit_distance
 * (src/compute-alignments.c:240), K_band_edit_distance (:319), general_refine_borders
 * (src/refine.c:105).  Static routines (find_longest_common_factor_dp, find_longest_affix)
 * cannot be interposed; they are exercised through oracle/ref_static_access.c instead.
 */
#include <dlfcn.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "types.h"
#include "list.h"

static FILE* out(void) {
  static FILE* f = NULL;
  static int tried = 0;
  if (!tried) {
    tried = 1;
    const char* p = getenv("PINTRON_DP_CAPTURE");
    if (p) f = fopen(p, "w");
  }
  return f;
}

static void put_str(FILE* f, const char* key, const char* s, size_t len) {
  fprintf(f, "\"%s\":\"", key);
  for (size_t i = 0; i < len; ++i) {
    unsigned char c = (unsigned char)s[i];
    if (c == '"' || c == '\\' || c < 0x20 || c > 0x7e) fprintf(f, "\\u%04x", c);
    else fputc(c, f);
  }
  fputc('"', f);
}

#define REAL(name) \
  static __typeof__(&name) real = NULL; \
  if (!real) real = (__typeof__(&name))dlsym(RTLD_NEXT, #name)

plist compute_alignment(char* est, char* gen, bool one);
plist compute_alignment(char* est, char* gen, bool one) {
  REAL(compute_alignment);
  plist r = real(est, gen, one);
  FILE* f = out();
  if (f) {
    palignment a = (palignment)list_head(r);
    fprintf(f, "{\"k\":\"ALIGN\",");
    put_str(f, "a", est, strlen(est)); fputc(',', f);
    put_str(f, "b", gen, strlen(gen));
    fprintf(f, ",\"score\":%d,\"dim\":%d,", a->score, a->alignment_dim);
    put_str(f, "ea", a->EST_alignment, strlen(a->EST_alignment)); fputc(',', f);
    put_str(f, "ga", a->GEN_alignment, strlen(a->GEN_alignment));
    fprintf(f, "}\n");
  }
  return r;
}

plist compute_gap_alignment(char* est, char* gen, bool one, int c, int cl, int cr);
plist compute_gap_alignment(char* est, char* gen, bool one, int c, int cl, int cr) {
  REAL(compute_gap_alignment);
  plist r = real(est, gen, one, c, cl, cr);
  FILE* f = out();
  if (f) {
    pgap_alignment a = (pgap_alignment)list_head(r);
    fprintf(f, "{\"k\":\"GAP\",");
    put_str(f, "a", est, strlen(est)); fputc(',', f);
    put_str(f, "b", gen, strlen(gen));
    fprintf(f, ",\"dim\":%d,\"factor_cut\":%d,\"intron_start\":%d,\"intron_end\":%d,"
               "\"intron_start_on_align\":%d,\"intron_end_on_align\":%d,",
            a->gap_alignment_dim, a->factor_cut, a->intron_start, a->intron_end,
            a->intron_start_on_align, a->intron_end_on_align);
    put_str(f, "ea", a->EST_gap_alignment, strlen(a->EST_gap_alignment)); fputc(',', f);
    put_str(f, "ga", a->GEN_gap_alignment, strlen(a->GEN_gap_alignment));
    fprintf(f, "}\n");
  }
  return r;
}

unsigned int* edit_distance(const char* const s1, const size_t ls1, const char* const s2, const size_t ls2);
unsigned int* edit_distance(const char* const s1, const size_t ls1, const char* const s2, const size_t ls2) {
  REAL(edit_distance);
  unsigned int* M = real(s1, ls1, s2, ls2);
  FILE* f = out();
  if (f) {
    fprintf(f, "{\"k\":\"ED\",");
    put_str(f, "a", s1, ls1); fputc(',', f);
    put_str(f, "b", s2, ls2);
    fprintf(f, ",\"score\":%u}\n", M[(ls1 + 1) * (ls2 + 1) - 1]);
  }
  return M;
}

size_t compute_edit_distance(const char* const s1, const size_t l1, const char* const s2, const size_t l2);
size_t compute_edit_distance(const char* const s1, const size_t l1, const char* const s2, const size_t l2) {
  REAL(compute_edit_distance);
  size_t d = real(s1, l1, s2, l2);
  FILE* f = out();
  if (f) {
    fprintf(f, "{\"k\":\"EDM\",");
    put_str(f, "a", s1, l1); fputc(',', f);
    put_str(f, "b", s2, l2);
    fprintf(f, ",\"score\":%zu}\n", d);
  }
  return d;
}

bool K_band_edit_distance(char* s1, char* s2, unsigned int ub, unsigned int* edit);
bool K_band_edit_distance(char* s1, char* s2, unsigned int ub, unsigned int* edit) {
  REAL(K_band_edit_distance);
  bool ok = real(s1, s2, ub, edit);
  FILE* f = out();
  if (f) {
    fprintf(f, "{\"k\":\"KBAND\",");
    put_str(f, "a", s1, strlen(s1)); fputc(',', f);
    put_str(f, "b", s2, strlen(s2));
    fprintf(f, ",\"ub\":%u,\"edit\":%u,\"ok\":%d}\n", ub, *edit, ok ? 1 : 0);
  }
  return ok;
}

bool general_refine_borders(const char* const p, const size_t len_p, const size_t min_p_cut,
                            const size_t max_p_cut, const char* const t, const size_t len_t,
                            const unsigned int max_errs, size_t* op, size_t* ot1, size_t* ot2,
                            unsigned int* oed);
bool general_refine_borders(const char* const p, const size_t len_p, const size_t min_p_cut,
                            const size_t max_p_cut, const char* const t, const size_t len_t,
                            const unsigned int max_errs, size_t* op, size_t* ot1, size_t* ot2,
                            unsigned int* oed) {
  REAL(general_refine_borders);
  bool ok = real(p, len_p, min_p_cut, max_p_cut, t, len_t, max_errs, op, ot1, ot2, oed);
  FILE* f = out();
  if (f) {
    fprintf(f, "{\"k\":\"BORDERS\",");
    put_str(f, "a", p, len_p); fputc(',', f);
    /* two bytes past len_t are readable in every reference call site (t is a NUL-terminated
     * string or a window into one) and getBursetFrequency_adaptor may look at them */
    size_t tl = len_t; if (t[tl] != '\0') { ++tl; if (t[tl] != '\0') ++tl; }
    put_str(f, "b", t, len_t); fputc(',', f);
    put_str(f, "b_tail", t + len_t, tl - len_t);
    fprintf(f, ",\"min_cut\":%zu,\"max_cut\":%zu,\"max_errs\":%u,\"off_p\":%zu,\"off_t1\":%zu,"
               "\"off_t2\":%zu,\"ed\":%u,\"ok\":%d}\n",
            min_p_cut, max_p_cut, max_errs, *op, *ot1, *ot2, *oed, ok ? 1 : 0);
  }
  return ok;
}
