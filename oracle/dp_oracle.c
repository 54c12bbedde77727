/*
 * oracle/dp_oracle.c -- TEST INFRASTRUCTURE ONLY (see dp_oracle.h).
 *
 * CPU restatement of the est-fact dynamic programs.  Written from the behaviour of the
 * reference routines (cited per function, paths relative to /root/reference); data layout and
 * control flow are our own.  Checked against the compiled reference in tests/.
 */
#include "dp_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int is_n(char c) { return c == 'n' || c == 'N'; }
/* N-wildcard equality used by ALIGN, GAP and LCF (compute-alignments.c:116-119,
 * refine-intron.c:671,746, factorization-refinement.c:279-283) */
static inline int eq_wild(char a, char b) { return a == b || is_n(a) || is_n(b); }

static void reverse_in_place(char* s, int32_t len) {
  for (int32_t a = 0, b = len - 1; a < b; ++a, --b) {
    char t = s[a]; s[a] = s[b]; s[b] = t;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* ALIGN                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* ComputeAlignMatrix (compute-alignments.c:85-147): unit-cost global alignment, wildcard N,
 * preference diagonal < up (EST char vs '-') < left ('-' vs genomic char) through strict '>'. */
uint32_t orc_align_matrix(const char* est, size_t n, const char* gen, size_t m, char* dir) {
  uint32_t* row = (uint32_t*)malloc((m + 1) * sizeof(uint32_t));
  for (size_t j = 0; j <= m; ++j) row[j] = (uint32_t)j;
  for (size_t i = 1; i <= n; ++i) {
    uint32_t diag = row[0];
    row[0] = (uint32_t)i;
    for (size_t j = 1; j <= m; ++j) {
      const uint32_t up = row[j];
      uint32_t best = diag + (eq_wild(est[i - 1], gen[j - 1]) ? 0u : 1u);
      char d = 0;
      if (best > up + 1) { best = up + 1; d = 1; }
      if (best > row[j - 1] + 1) { best = row[j - 1] + 1; d = 2; }
      dir[i * m + j] = d;          /* row stride m, not m+1 (compute-alignments.c:135) */
      diag = up;
      row[j] = best;
    }
  }
  const uint32_t score = row[m];
  free(row);
  return score;
}

uint32_t orc_align(const char* est, size_t n, const char* gen, size_t m,
                   char* est_aln, char* gen_aln, int32_t* dim) {
  /* equal-string shortcut (compute-alignments.c:48-58) */
  if (est == gen || (n == m && memcmp(est, gen, n) == 0)) {
    memcpy(est_aln, est, n); est_aln[n] = '\0';
    memcpy(gen_aln, gen, n); gen_aln[n] = '\0';
    *dim = (int32_t)n;
    return 0;
  }
  char* dir = (char*)calloc((n + 1) * (m + 1), 1);
  const uint32_t score = orc_align_matrix(est, n, gen, m, dir);
  /* TracebackAlignment (compute-alignments.c:149-207) */
  int32_t k = 0;
  size_t i = n, j = m;
  while (i > 0 && j > 0) {
    const char d = dir[i * m + j];
    if (d == 0)      { est_aln[k] = est[--i]; gen_aln[k] = gen[--j]; }
    else if (d == 1) { est_aln[k] = est[--i]; gen_aln[k] = '-'; }
    else             { est_aln[k] = '-';      gen_aln[k] = gen[--j]; }
    ++k;
  }
  while (i > 0) { est_aln[k] = est[--i]; gen_aln[k] = '-'; ++k; }
  while (j > 0) { est_aln[k] = '-'; gen_aln[k] = gen[--j]; ++k; }
  est_aln[k] = gen_aln[k] = '\0';
  reverse_in_place(est_aln, k);
  reverse_in_place(gen_aln, k);
  *dim = k;
  free(dir);
  return score;
}

uint64_t orc_cells_align(const char* est, size_t n, const char* gen, size_t m) {
  if (est == gen || (n == m && memcmp(est, gen, n) == 0)) return 0;
  return (uint64_t)n * m;
}

/* ------------------------------------------------------------------------------------------ */
/* Levenshtein (no wildcard)                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* edit_distance (refine.c:50-83): rows follow s2, columns follow s1. */
void orc_edit_distance_full(const char* s1, size_t ls1, const char* s2, size_t ls2, uint32_t* M) {
  const size_t w = ls1 + 1;
  for (size_t c = 0; c <= ls1; ++c) M[c] = (uint32_t)c;
  for (size_t r = 1; r <= ls2; ++r) {
    uint32_t* cur = M + r * w;
    const uint32_t* prev = cur - w;
    cur[0] = (uint32_t)r;
    for (size_t c = 1; c <= ls1; ++c) {
      uint32_t v = prev[c - 1] + (s2[r - 1] == s1[c - 1] ? 0u : 1u);
      if (prev[c] + 1 < v) v = prev[c] + 1;
      if (cur[c - 1] + 1 < v) v = cur[c - 1] + 1;
      cur[c] = v;
    }
  }
}

uint32_t orc_edit_distance(const char* a, size_t la, const char* b, size_t lb) {
  uint32_t* row = (uint32_t*)malloc((lb + 1) * sizeof(uint32_t));
  for (size_t j = 0; j <= lb; ++j) row[j] = (uint32_t)j;
  for (size_t i = 1; i <= la; ++i) {
    uint32_t diag = row[0];
    row[0] = (uint32_t)i;
    for (size_t j = 1; j <= lb; ++j) {
      const uint32_t up = row[j];
      uint32_t v = diag + (a[i - 1] == b[j - 1] ? 0u : 1u);
      if (up + 1 < v) v = up + 1;
      if (row[j - 1] + 1 < v) v = row[j - 1] + 1;
      diag = up;
      row[j] = v;
    }
  }
  const uint32_t d = row[lb];
  free(row);
  return d;
}

/* compute_best_suffix_cut (compute-alignments.c:252-292): minima of the last column / last row
 * of edit_distance_matrix(s1,s2) (rows s1, cols s2); `>=` moves ties to the LARGEST index below
 * the corner; the corner itself is the starting value. */
uint32_t orc_best_suffix_cut(const char* s1, size_t l1, const char* s2, size_t l2,
                             uint32_t* cut1, uint32_t* cut2) {
  if (l1 == l2 && strncmp(s1, s2, l1) == 0) {
    *cut1 = (uint32_t)l1; *cut2 = (uint32_t)l2;
    return 0;
  }
  /* matrix[r][c], r over s1, c over s2 == transposed layout of orc_edit_distance_full(s2,s1) */
  uint32_t* M = (uint32_t*)malloc((l1 + 1) * (l2 + 1) * sizeof(uint32_t));
  orc_edit_distance_full(s2, l2, s1, l1, M);
  const size_t w = l2 + 1;
  uint32_t mincol = M[l1 * w + l2], minrow = mincol;
  size_t mincolpos = l1, minrowpos = l2;
  for (size_t r = 0; r < l1; ++r)
    if (mincol >= M[r * w + l2]) { mincol = M[r * w + l2]; mincolpos = r; }
  for (size_t c = 0; c < l2; ++c)
    if (minrow >= M[l1 * w + c]) { minrow = M[l1 * w + c]; minrowpos = c; }
  free(M);
  if (minrow < mincol) { *cut1 = (uint32_t)l1; *cut2 = (uint32_t)minrowpos; return minrow; }
  *cut1 = (uint32_t)mincolpos; *cut2 = (uint32_t)l2;
  return mincol;
}

uint32_t orc_best_prefix_cut(const char* s1, size_t l1, const char* s2, size_t l2,
                             uint32_t* cut1, uint32_t* cut2) {
  if (l1 == l2 && strncmp(s1, s2, l1) == 0) { *cut1 = 0; *cut2 = 0; return 0; }
  char* r1 = (char*)malloc(l1 + 1);
  char* r2 = (char*)malloc(l2 + 1);
  for (size_t i = 0; i < l1; ++i) r1[i] = s1[l1 - 1 - i];
  for (size_t i = 0; i < l2; ++i) r2[i] = s2[l2 - 1 - i];
  r1[l1] = r2[l2] = '\0';
  const uint32_t ed = orc_best_suffix_cut(r1, l1, r2, l2, cut1, cut2);
  *cut1 = (uint32_t)l1 - *cut1;
  *cut2 = (uint32_t)l2 - *cut2;
  free(r1); free(r2);
  return ed;
}

/* ------------------------------------------------------------------------------------------ */
/* KBAND                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* K_band_edit_distance (compute-alignments.c:319-453).
 * Band cell (r, c) -- r over the shorter string (1..m), c over the longer one (1..n) -- is kept
 * at band slot c - r + k, slots 0..2k.  Slots outside the matrix hold the sentinel k+1.  The
 * reference keeps two rolling rows and never re-initialises them, so a slot that a row does not
 * write still holds what was written two rows before; we keep that behaviour by using two rows
 * and writing exactly the slots the reference writes. */
/* getDinucleotideIndex (src/exon-complexity.c:80-130): A, C, G, T in either case -> 4 * first + second, anything else 16 */
static int dust_base(char c) {
  switch (c) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; default: return -1; }
}
/* dustScore (src/exon-complexity.c:50-78): every dinucleotide adds the number of times it has been seen before;
 * 10 x that sum / (length - 2), then / length; 0 for fewer than three characters */
double orc_dust_score(const char* s, size_t len) {
  if ((int)len <= 2) return 0.0;
  int freq[17] = { 0 };
  int running = 0;
  for (int i = 0; i < (int)len - 1; ++i) {
    const int a = dust_base(s[i]), b = dust_base(s[i + 1]);
    const int index = (a < 0 || b < 0) ? 16 : 4 * a + b;
    running += freq[index];
    freq[index]++;
  }
  const double dust = (10.0 * (double)running) / ((double)(len - 2));
  return dust / len;
}
uint32_t orc_dust_flags(const char* gen, size_t lg, const char* est, size_t le, double threshold) {
  return (orc_dust_score(gen, lg) > threshold ? 1u : 0u) | (orc_dust_score(est, le) > threshold ? 2u : 0u);
}

int orc_kband(const char* seq1, size_t l1, const char* seq2, size_t l2, uint32_t upper_bound,
              uint32_t* edit) {
  if (l1 == l2 && memcmp(seq1, seq2, l1) == 0) { *edit = 0; return 1; }
  if (upper_bound == 0) { *edit = 1; return 0; }
  const char* lng = seq1; const char* sht = seq2;
  size_t n = l1, m = l2;
  if (l1 < l2) { lng = seq2; sht = seq1; n = l2; m = l1; }
  if (n - m > upper_bound) { *edit = (uint32_t)(n - m); return 0; }
  const size_t k = upper_bound;
  if (2 * k + 1 >= n) {
    /* compute_edit_distance(seq1,n,seq2,m) incl. its own equal-string shortcut (:240-249) */
    *edit = (n == m && memcmp(lng, sht, n) == 0) ? 0 : orc_edit_distance(lng, n, sht, m);
    return *edit <= upper_bound;
  }
  const size_t W = 2 * k + 1;
  uint64_t* A = (uint64_t*)malloc(W * sizeof(uint64_t));   /* "M1": previous row */
  uint64_t* B = (uint64_t*)malloc(W * sizeof(uint64_t));   /* "M2": row being written */
  /* M1[0..k-1] is never initialised by the reference before row 1 reads M1[k-1+c]... those reads
   * start at slot k (c=1 reads M1[k-1+1]), so slots below k of the first row are not read. */
  for (size_t c = 0; c < W; ++c) A[c] = 0;
  for (size_t c = 0; c <= k; ++c) A[k + c] = c;
  for (size_t c = 0; c < W; ++c) B[c] = k + 1;
  uint64_t d;
#define MIN2(x, y) ((x) < (y) ? (x) : (y))
  for (size_t r = 1; r <= k; ++r) {                       /* rows whose band sticks out left */
    B[k - r] = r;
    for (size_t c = 1; c < r + k; ++c) {
      d = A[k - r + c] + (lng[c - 1] != sht[r - 1]);
      d = MIN2(d, B[k - r + c - 1] + 1);
      d = MIN2(d, A[k - r + c + 1] + 1);
      B[k - r + c] = d;
    }
    d = A[2 * k] + (lng[r + k - 1] != sht[r - 1]);
    d = MIN2(d, B[2 * k - 1] + 1);
    B[2 * k] = d;
    uint64_t* t = A; A = B; B = t;
  }
  for (size_t r = k + 1; r <= n - k; ++r) {                /* full-width rows */
    d = A[0] + (lng[r - k - 1] != sht[r - 1]);
    B[0] = MIN2(d, A[1] + 1);
    for (size_t c = r + 1 - k; c < r + k; ++c) {
      d = A[c + k - r] + (lng[c - 1] != sht[r - 1]);
      d = MIN2(d, B[c + k - r - 1] + 1);
      d = MIN2(d, A[c + k - r + 1] + 1);
      B[c + k - r] = d;
    }
    d = A[2 * k] + (lng[r + k - 1] != sht[r - 1]);
    d = MIN2(d, B[2 * k - 1] + 1);
    B[2 * k] = d;
    uint64_t* t = A; A = B; B = t;
  }
  for (size_t r = n + 1 - k; r <= m; ++r) {                /* rows whose band sticks out right */
    d = A[0] + (lng[r - k - 1] != sht[r - 1]);
    B[0] = MIN2(d, A[1] + 1);
    for (size_t c = r + 1 - k; c <= n; ++c) {
      d = A[c + k - r] + (lng[c - 1] != sht[r - 1]);
      d = MIN2(d, B[c + k - r - 1] + 1);
      d = MIN2(d, A[c + k - r + 1] + 1);
      B[c + k - r] = d;
    }
    uint64_t* t = A; A = B; B = t;
  }
#undef MIN2
  const uint64_t result = A[n + k - m];
  free(A); free(B);
  *edit = (uint32_t)result;
  return result <= upper_bound;
}

uint64_t orc_cells_kband(const char* s1, size_t l1, const char* s2, size_t l2, uint32_t k) {
  if (l1 == l2 && memcmp(s1, s2, l1) == 0) return 0;
  if (k == 0) return 0;
  size_t n = l1 > l2 ? l1 : l2, m = l1 > l2 ? l2 : l1;
  if (n - m > k) return 0;
  if (2 * (size_t)k + 1 >= n) return (uint64_t)n * m;
  return (uint64_t)m * (2 * (uint64_t)k + 1);
}

/* ------------------------------------------------------------------------------------------ */
/* GAP                                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* ComputeGapAlignMatrix with only_one_align (refine-intron.c:623-824): three score planes over
 * (n+1)x(m+1), everything (borders included) starts at 0.
 *   L: exon left of the intron    diag +-1, up -1, left -1           dirs 0 / 1 / 2
 *   G: the intron (cost-free run of genomic characters)  stay in G (2) or enter from L (-2)
 *   R: exon right of the intron   diag +-1, left -1 (0 in the last EST row), from G (-2), up -1
 * every alternative replaces the running best only when strictly larger, in the listed order. */
void orc_gap_align(const char* est, size_t n, const char* gen, size_t m,
                   char* est_aln, char* gen_aln, orc_gap_result* res) {
  const size_t w = m + 1, cells = (n + 1) * w;
  int32_t* L = (int32_t*)calloc(cells, sizeof(int32_t));
  int32_t* G = (int32_t*)calloc(cells, sizeof(int32_t));
  int32_t* R = (int32_t*)calloc(cells, sizeof(int32_t));
  signed char* dL = (signed char*)calloc(cells, 1);
  signed char* dG = (signed char*)calloc(cells, 1);
  signed char* dR = (signed char*)calloc(cells, 1);

  for (size_t i = 1; i <= n; ++i) {
    for (size_t j = 1; j <= m; ++j) {
      const size_t x = i * w + j;
      const int32_t sub = eq_wild(est[i - 1], gen[j - 1]) ? 1 : -1;
      /* L */
      int32_t v = L[x - w - 1] + sub; signed char d = 0;
      if (v < L[x - w] - 1) { v = L[x - w] - 1; d = 1; }
      if (v < L[x - 1] - 1) { v = L[x - 1] - 1; d = 2; }
      L[x] = v; dL[x] = d;
      /* G  (depends on L of the previous column only) */
      v = G[x - 1]; d = 2;
      if (v < L[x - 1]) { v = L[x - 1]; d = -2; }
      G[x] = v; dG[x] = d;
      /* R  (depends on G of the previous column only) */
      v = R[x - w - 1] + sub; d = 0;
      const int32_t left = (i != n) ? R[x - 1] - 1 : R[x - 1];   /* free trailing gap :756-759 */
      if (v < left) { v = left; d = 2; }
      if (v < G[x - 1]) { v = G[x - 1]; d = -2; }
      if (v < R[x - w] - 1) { v = R[x - w] - 1; d = 1; }
      R[x] = v; dR[x] = d;
    }
  }
  const size_t last = n * w + m;
  int32_t plane;
  if (R[last] >= G[last]) plane = (R[last] >= L[last]) ? 2 : 0;      /* :808-819 */
  else                    plane = (G[last] >= L[last]) ? 1 : 0;
  res->start_matrix = plane;
  res->score = plane == 2 ? R[last] : (plane == 1 ? G[last] : L[last]);
  res->factor_cut = res->intron_start = res->intron_end = 0;
  res->intron_start_on_align = res->intron_end_on_align = 0;

  /* TracebackGapAlignment (:828-890) is recursive and emits columns on the way back; we walk
   * backwards, record the reversed column index of each jump and fix the indices afterwards. */
  int32_t k = 0;
  int32_t rev_end = -1, rev_start = -1;     /* reversed positions of the two jump columns */
  size_t i = n, j = m;
  while (i > 0 && j > 0) {
    const size_t x = i * w + j;
    const signed char d = plane == 2 ? dR[x] : (plane == 1 ? dG[x] : dL[x]);
    if (d == 0)      { est_aln[k] = est[--i]; gen_aln[k] = gen[--j]; }
    else if (d == 1) { est_aln[k] = est[--i]; gen_aln[k] = '-'; }
    else {
      if (d == -2) {
        if (plane == 2) { res->intron_end = (int32_t)j - 1; res->factor_cut = (int32_t)i; rev_end = k; }
        else            { res->intron_start = (int32_t)j - 1; rev_start = k; }
        --plane;
      }
      est_aln[k] = '-'; gen_aln[k] = gen[--j];
    }
    ++k;
  }
  while (i > 0) { est_aln[k] = est[--i]; gen_aln[k] = '-'; ++k; }
  while (j > 0) { est_aln[k] = '-'; gen_aln[k] = gen[--j]; ++k; }
  est_aln[k] = gen_aln[k] = '\0';
  reverse_in_place(est_aln, k);
  reverse_in_place(gen_aln, k);
  if (rev_end >= 0)   res->intron_end_on_align = k - 1 - rev_end;
  if (rev_start >= 0) res->intron_start_on_align = k - 1 - rev_start;
  res->dim = k;
  free(L); free(G); free(R); free(dL); free(dG); free(dR);
}

/* ------------------------------------------------------------------------------------------ */
/* LCF                                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* find_longest_common_factor_dp (factorization-refinement.c:255-316): longest run of wildcard
 * matches along a diagonal; strict '<' keeps the FIRST maximum in (i1, i2) scan order.  The
 * swapped-argument recursion at :260-262 is followed by the un-swapped pass, which overwrites
 * its outputs, so only the un-swapped pass is observable. */
void orc_lcf(const char* s1, size_t l1, const char* s2, size_t l2,
             uint32_t* occ1, uint32_t* occ2, uint32_t* len) {
  uint32_t* run = (uint32_t*)calloc(l2 + 1, sizeof(uint32_t));
  uint32_t best = 0, b1 = 0, b2 = 0;
  for (size_t i1 = 0; i1 < l1; ++i1) {
    /* walk i2 downwards so run[i2] still holds the previous row's diagonal neighbour;
     * the reference walks upwards with two rows -- ties must still resolve to the smallest i2,
     * hence the `<=` on equal length within a row. */
    uint32_t row_best = 0, row_b2 = 0;
    for (size_t i2 = l2; i2-- > 0;) {
      const uint32_t v = eq_wild(s1[i1], s2[i2]) ? run[i2] + 1 : 0;
      run[i2 + 1] = v;
      if (v >= row_best && v > 0) { row_best = v; row_b2 = (uint32_t)(i2 + 1 - v); }
    }
    run[0] = 0;
    if (row_best > best) { best = row_best; b1 = (uint32_t)(i1 + 1 - row_best); b2 = row_b2; }
  }
  free(run);
  *occ1 = b1; *occ2 = b2; *len = best;
}

/* ------------------------------------------------------------------------------------------ */
/* Burset frequencies                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* getBursetFrequency (refine-intron.c:376-556) as data: index = 4 bases (donor[0], donor[1],
 * acceptor[0], acceptor[1]) at 2 bits each, A=0 C=1 G=2 T=3; 58 non-zero pairs. */
static const unsigned char burset_freq[256] = {
    0,   0,   1,   1,   0,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,
    0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   1,   5,   0,   0,   0,   0,   2,   0,   1,   0,   0,   0,   0,   2,   0,
    1,   8,   7,   2,   0,   0,   0,   0,   0,   1,   0,   1,   0,   0,   0,   0,
    0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   1,
    0,   0,   2,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   0,   1,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   2,   0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   0,   8,   0,   0,   0,   0,   0,   0,   0,   0,   1,   0,   1,   1,   0,
    0,   0, 126,   0,   0,   0,   0,   0,   0,   0,   1,   0,   1,   0,   0,   0,
    0,   1,  11,   0,   1,   0,   0,   0,   2,   0,   0,   0,   0,   2,   0,   0,
    0,   4, 200,   2,   9,   0,   4,   3,   0,   1,  10,   1,   7,   2,   8,   2,
    0,   0,   6,   0,   0,   0,   1,   0,   0,   0,   0,   0,   0,   1,   0,   0,
    0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,
    0,   1,   7,   0,   0,   0,   0,   0,   0,   0,   2,   0,   0,   0,   0,   0,
    0,   0,   5,   1,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,
};

static int base_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

static int burset4(char d0, char d1, char a0, char a1) {
  const int c0 = base_code(d0), c1 = base_code(d1), c2 = base_code(a0), c3 = base_code(a1);
  if ((c0 | c1 | c2 | c3) < 0) return 0;
  return burset_freq[(c0 << 6) | (c1 << 4) | (c2 << 2) | c3];
}

int orc_burset_frequency(const char* donor, const char* acceptor) {
  /* the reference compares whole strings: anything that is not exactly two bases scores 0 */
  if (strlen(donor) != 2 || strlen(acceptor) != 2) return 0;
  return burset4(donor[0], donor[1], acceptor[0], acceptor[1]);
}

int orc_burset_adaptor(const char* t, size_t cut1, size_t cut2) {
  if (cut2 < 2) return 0;
  /* a NUL inside either dinucleotide shortens the reference's C string => no match */
  if (t[cut1] == '\0' || t[cut1 + 1] == '\0' || t[cut2 - 2] == '\0' || t[cut2 - 1] == '\0') return 0;
  return burset4(t[cut1], t[cut1 + 1], t[cut2 - 2], t[cut2 - 1]);
}

/* ------------------------------------------------------------------------------------------ */
/* general_refine_borders                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* row minima of edit_distance(text[0..t_win), pat) with first arg-min (refine.c:128-159) */
static void row_minima(const char* text, size_t t_win, const char* pat, size_t len_p,
                       uint32_t* minv, uint32_t* minpos) {
  uint32_t* M = (uint32_t*)malloc((t_win + 1) * (len_p + 1) * sizeof(uint32_t));
  orc_edit_distance_full(text, t_win, pat, len_p, M);
  minv[0] = 0; minpos[0] = 0;
  for (size_t i = 1; i <= len_p; ++i) {
    const uint32_t* row = M + i * (t_win + 1);
    uint32_t bv = row[0], bp = 0;
    for (size_t j = 1; j <= t_win; ++j)
      if (bv > row[j]) { bv = row[j]; bp = (uint32_t)j; }
    minv[i] = bv; minpos[i] = bp;
  }
  free(M);
}

void orc_refine_borders(const char* p, size_t len_p, size_t min_p_cut, size_t max_p_cut,
                        const char* t, size_t len_t, uint32_t max_errs,
                        orc_borders_result* res) {
  const size_t t_win = (len_p + max_errs < len_t) ? len_p + max_errs : len_t;
  char* rt = (char*)malloc(len_t + 1);
  char* rp = (char*)malloc(len_p + 1);
  for (size_t i = 0; i < len_t; ++i) rt[i] = t[len_t - 1 - i];
  for (size_t i = 0; i < len_p; ++i) rp[i] = p[len_p - 1 - i];
  rt[len_t] = rp[len_p] = '\0';
  uint32_t* pre = (uint32_t*)malloc(4 * (len_p + 1) * sizeof(uint32_t));
  uint32_t* pre_pos = pre + (len_p + 1);
  uint32_t* suf = pre_pos + (len_p + 1);
  uint32_t* suf_pos = suf + (len_p + 1);
  row_minima(t, t_win, p, len_p, pre, pre_pos);
  row_minima(rt, t_win, rp, len_p, suf, suf_pos);

  size_t off_p = min_p_cut;
  size_t off_t1 = pre_pos[min_p_cut];
  size_t off_t2 = suf_pos[len_p - min_p_cut];
  uint32_t best = pre[min_p_cut] + suf[len_p - min_p_cut];
  int best_freq = orc_burset_adaptor(t, off_t1, len_t - off_t2);
  for (size_t i = min_p_cut + 1; i <= max_p_cut; ++i) {
    const int freq = orc_burset_adaptor(t, pre_pos[i], len_t - suf_pos[len_p - i]);
    const uint32_t cur = pre[i] + suf[len_p - i];
    if (best > cur || (best == cur && freq > best_freq)) {
      best = cur; off_p = i; off_t1 = pre_pos[i]; off_t2 = suf_pos[len_p - i]; best_freq = freq;
    }
  }
  res->offset_p = (uint32_t)off_p;
  res->offset_t1 = (uint32_t)off_t1;
  res->offset_t2 = (uint32_t)(len_t - off_t2);
  res->edit_distance = best;
  res->ok = best <= max_errs;
  free(pre); free(rt); free(rp);
}

/* ------------------------------------------------------------------------------------------ */
/* find_longest_affix                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* factorization-refinement.c:1136-1173: scan every (ecut,gcut) >= 1 in row-major order; a cell
 * qualifies when the cut characters match and weight = 2*ed/(ecut+gcut) <= 0.17 (IEEE double);
 * `<=` against the running best lets later equal-weight cells win. */
int orc_longest_affix(const char* est, size_t estl, const char* gen, size_t genl,
                      uint32_t* ecut, uint32_t* gcut) {
  uint32_t* M = (uint32_t*)malloc((estl + 1) * (genl + 1) * sizeof(uint32_t));
  /* edit_distance_matrix(est, gen): rows est, cols gen == orc_edit_distance_full(gen, est) */
  orc_edit_distance_full(gen, genl, est, estl, M);
  int valid = 0;
  double best_w = 1.0;
  uint32_t be = 0, bg = 0;
  for (size_t e = 1; e <= estl; ++e)
    for (size_t g = 1; g <= genl; ++g) {
      const double wgt = 2.0 * ((double)M[e * (genl + 1) + g]) / (double)(e + g);
      if (est[e - 1] == gen[g - 1] && wgt <= 0.17 && wgt <= best_w) {
        be = (uint32_t)e; bg = (uint32_t)g; best_w = wgt; valid = 1;
      }
    }
  free(M);
  if (valid) { *ecut = be; *gcut = bg; }
  return valid;
}
