/*
 * oracle/pairing_oracle.c -- TEST INFRASTRUCTURE ONLY (see pairing_oracle.h).
 *
 * What the reference computes (src/max-emb-graph.c:218-392, src/aug_suffix_tree.c:150-245), said
 * without the tree.  T = genomic, P = pattern, L = min_factor_len, for every position i of P:
 *
 *  - an occurrence t of P[i..] is "prev-excluded" when i > 0, t > 0 and T[t-1] == P[i-1]
 *    (fill_list_pairings skips the slice of the preceding symbol, :178-181; t == 0 is kept, :195);
 *  - A_i = the longest match of P[i..] among the occurrences that are not prev-excluded
 *    (the descent prunes a child only when its whole subtree is prev-excluded, :77-85);
 *  - the descent does not start at the root but at the suffix link of the previous locus
 *    (:254-262,143-163), i.e. at depth s_i, and only prunes below that depth, so the locus depth
 *    is D_i = max(A_i, s_i);  s_{i+1} = D_i - 1 when the locus is an explicit node, else
 *    (depth of the explicit node above it) - 1, and 0 when that node is the root or nothing
 *    matched (:250-253, prev_N == NULL);
 *  - threshold thr_i = (size_t) max(D_i * rate, L) evaluated in double (:273-274);
 *  - pairings (i, t, l): every occurrence that is not prev-excluded, l = lcp(P[i..], T[t..]),
 *    l >= thr_i (walk towards the root, :281-300), sorted by (t, l) (:301);
 *  - filter (a) inside a position (:302-334), filter (b) across adjacent positions (:349-375).
 *
 * Explicit nodes are read off the LCP array: the SA interval [lo,hi] of P[i..i+D) lies below the
 * explicit node of depth max(lcp[lo], lcp[hi+1]); the locus itself is an explicit node iff the
 * first and last suffix of the interval differ at offset D.
 * Depths below L are not tracked exactly (they cannot influence a threshold or a pairing).
 *
 * The occurrence t == 0 has no preceding character: the reference files it under EVERY symbol of the
 * genomic alphabet (src/aug_suffix_tree.c:183-192) and fill_list_pairings reports it once per
 * symbol slice it walks, except where a guard stops it (:195: only in the slice of key 0, or key 1
 * when key 0 is the slice of the preceding symbol).  The guard sits in the loop over the entries
 * BEFORE the block of the child already reported, not in the loop over the entries AFTER it
 * (:203-211) -- and suffix 0 is always after: libstree adds children at the head of the child list
 * (stree_src/lst_stree.c:87) and the branch towards suffix 0 is the oldest of every node on its
 * path (it is the remainder of the edge that was split, :548-590), so it comes last in the
 * depth-first order that fills the occurrence array.  Hence, with l0 = lcp(P[i..], T[0..]) >= thr_i:
 *     copies of (i, 0, l0) = sum over keys k != key(P[i-1]) of
 *                              1            when some occurrence t' > 0 that matches MORE than l0
 *                                           characters and is not prev-excluded has key(T[t'-1]) == k
 *                                           (the block's slice k is not empty: second loop),
 *                              guard(k)     otherwise (first loop),
 *     guard(k) = (k == 0 || (k == 1 && key(P[i-1]) == 0)); key(c) = rank of c among the distinct
 *     characters of T, or their number when c does not occur in T (or i == 0).
 * Pinned against the reference's own build_vertex_set in tests/test_pairings_oracle.py.
 */
#define _GNU_SOURCE
#include "pairing_oracle.h"

#include <stdlib.h>
#include <string.h>

struct orc_index {
  const char* T;        /* private copy */
  size_t n;
  uint32_t* sa;
  uint32_t* lcp;        /* lcp[k] = lcp(suffix sa[k-1], suffix sa[k]); lcp[0] = lcp[n] = 0 */
  unsigned char key[256];  /* preprocess_text (src/aug_suffix_tree.c:68-120): rank among T's distinct characters */
  unsigned sigma;          /* their number = the key of characters that do not occur */
};

static int suffix_cmp(const void* a, const void* b, void* ctx) {
  const struct orc_index* ix = (const struct orc_index*)ctx;
  const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  const size_t lx = ix->n - x, ly = ix->n - y;
  const int c = memcmp(ix->T + x, ix->T + y, lx < ly ? lx : ly);
  if (c) return c;
  return lx < ly ? -1 : (lx > ly ? 1 : 0);      /* a proper prefix sorts first (terminator) */
}

orc_index* orc_index_create(const char* genomic, size_t n) {
  orc_index* ix = (orc_index*)calloc(1, sizeof(*ix));
  char* t = (char*)malloc(n + 1);
  memcpy(t, genomic, n); t[n] = '\0';
  ix->T = t; ix->n = n;
  {
    char seen[256] = {0};
    for (size_t i = 0; i < n; ++i) seen[(unsigned char)t[i]] = 1;
    for (int c = 0; c < 256; ++c) if (seen[c]) ix->key[c] = (unsigned char)ix->sigma++;
    for (int c = 0; c < 256; ++c) if (!seen[c]) ix->key[c] = (unsigned char)ix->sigma;
  }
  ix->sa = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
  ix->lcp = (uint32_t*)calloc(n + 2, sizeof(uint32_t));
  for (size_t i = 0; i < n; ++i) ix->sa[i] = (uint32_t)i;
  qsort_r(ix->sa, n, sizeof(uint32_t), suffix_cmp, ix);
  /* Kasai */
  uint32_t* rank = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
  for (size_t k = 0; k < n; ++k) rank[ix->sa[k]] = (uint32_t)k;
  size_t h = 0;
  for (size_t i = 0; i < n; ++i) {
    if (rank[i] == 0) { h = 0; continue; }
    const size_t j = ix->sa[rank[i] - 1];
    while (i + h < n && j + h < n && t[i + h] == t[j + h]) ++h;
    ix->lcp[rank[i]] = (uint32_t)h;
    if (h) --h;
  }
  free(rank);
  return ix;
}

void orc_index_destroy(orc_index* ix) {
  if (!ix) return;
  free((void*)ix->T); free(ix->sa); free(ix->lcp); free(ix);
}
const uint32_t* orc_index_sa(const orc_index* ix) { return ix->sa; }
const uint32_t* orc_index_lcp(const orc_index* ix) { return ix->lcp; }

/* compare suffix t of T against the d-character string q: <0, 0 (q is a prefix of the suffix), >0 */
static int cmp_prefix(const orc_index* ix, uint32_t t, const char* q, size_t d) {
  const size_t avail = ix->n - t;
  const size_t c = avail < d ? avail : d;
  const int r = memcmp(ix->T + t, q, c);
  if (r) return r;
  return avail < d ? -1 : 0;
}

/* SA interval [lo,hi) of suffixes having q[0..d) as a prefix, searched inside [from,to) */
static void interval(const orc_index* ix, const char* q, size_t d, size_t from, size_t to,
                     size_t* lo, size_t* hi) {
  size_t a = from, b = to;
  while (a < b) { const size_t mid = (a + b) / 2; if (cmp_prefix(ix, ix->sa[mid], q, d) < 0) a = mid + 1; else b = mid; }
  *lo = a;
  b = to;
  while (a < b) { const size_t mid = (a + b) / 2; if (cmp_prefix(ix, ix->sa[mid], q, d) <= 0) a = mid + 1; else b = mid; }
  *hi = a;
}

typedef struct { int32_t t, l; } occ_t;

static int occ_cmp(const void* a, const void* b) {
  const occ_t* x = (const occ_t*)a; const occ_t* y = (const occ_t*)b;
  if (x->t != y->t) return x->t < y->t ? -1 : 1;
  return x->l - y->l;
}

long orc_pairings(const orc_index* ix, const char* P, size_t m, uint32_t L, double rate,
                  int32_t* out, long cap, int32_t* depth_out) {
  const char* T = ix->T;
  const size_t n = ix->n;
  /* per-position lists after filter (a) */
  occ_t** lists = (occ_t**)calloc(m + 1, sizeof(occ_t*));
  size_t* cnt = (size_t*)calloc(m + 1, sizeof(size_t));
  size_t s = 0;                                   /* start depth of the descent */
  for (size_t i = 0; i < m; ++i) {
    size_t D = 0;
    size_t lo = 0, hi = 0;
    occ_t* occ = NULL; size_t nocc = 0;
    size_t A = 0;
    if (m - i >= L && L > 0) {
      interval(ix, P + i, L, 0, n, &lo, &hi);
      occ = (occ_t*)malloc((hi - lo + 1) * sizeof(occ_t));
      for (size_t k = lo; k < hi; ++k) {
        const size_t t = ix->sa[k];
        if (i > 0 && t > 0 && T[t - 1] == P[i - 1]) continue;          /* prev-excluded */
        size_t l = L;
        while (i + l < m && t + l < n && P[i + l] == T[t + l]) ++l;
        occ[nocc].t = (int32_t)t; occ[nocc].l = (int32_t)l; ++nocc;
        if (l > A) A = l;
      }
    }
    /* copies of the t == 0 occurrence (header comment): extra ones are appended, none may remain */
    for (size_t z = 0; z < nocc; ++z) {
      if (occ[z].t != 0) continue;
      const int32_t l0 = occ[z].l;
      const unsigned sk = i > 0 ? ix->key[(unsigned char)P[i - 1]] : ix->sigma;
      unsigned present = 0;
      for (size_t q = 0; q < nocc; ++q)
        if (occ[q].t > 0 && occ[q].l > l0) present |= 1u << ix->key[(unsigned char)T[occ[q].t - 1]];
      unsigned copies = 0;
      for (unsigned k = 0; k < ix->sigma; ++k) {
        if (k == sk) continue;
        copies += ((present >> k) & 1u) ? 1u : ((k == 0 || (k == 1 && sk == 0)) ? 1u : 0u);
      }
      if (copies == 0) { occ[z] = occ[--nocc]; }
      else if (copies > 1) {
        occ = (occ_t*)realloc(occ, (nocc + copies) * sizeof(occ_t));
        for (unsigned c = 1; c < copies; ++c) occ[nocc++] = occ[z];
      }
      break;
    }
    D = A > s ? A : s;
    if (D < L) {                                   /* below the tracked range: restart at the root */
      if (depth_out) depth_out[i] = 0;
      free(occ);
      s = 0;
      continue;
    }
    if (depth_out) depth_out[i] = (int32_t)D;
    const double thr_d = ((double)D * rate > (double)L) ? (double)D * rate : (double)L;
    const size_t thr = (size_t)thr_d;
    size_t keep = 0;
    for (size_t k = 0; k < nocc; ++k) if ((size_t)occ[k].l >= thr) occ[keep++] = occ[k];
    qsort(occ, keep, sizeof(occ_t), occ_cmp);
    /* filter (a), :302-334: drop PJ when an earlier PI covers it or is its shifted twin */
    char* dead = (char*)calloc(keep + 1, 1);
    for (size_t j = keep; j-- > 1;) {
      for (size_t q = j; q-- > 0;) {
        const occ_t* PI = &occ[q]; const occ_t* PJ = &occ[j];
        if ((PJ->t > PI->t && PJ->t + PJ->l <= PI->t + PI->l) ||
            (PJ->t == PI->t + 1 && PJ->l == PI->l)) { dead[j] = 1; break; }
      }
    }
    size_t w = 0;
    for (size_t k = 0; k < keep; ++k) if (!dead[k]) occ[w++] = occ[k];
    free(dead);
    lists[i] = occ; cnt[i] = w;
    /* next start depth from the locus at depth D on the path of P[i..] */
    size_t l2, h2;
    if (D == L) { l2 = lo; h2 = hi; } else interval(ix, P + i, D, lo, hi, &l2, &h2);
    const size_t par = ix->lcp[l2] > ix->lcp[h2] ? ix->lcp[l2] : ix->lcp[h2];   /* lcp[h2] = border after the interval */
    if (par == 0) s = 0;
    else {
      int at_node = 0;
      if (h2 - l2 >= 2) {
        const size_t t1 = ix->sa[l2], t2 = ix->sa[h2 - 1];
        const int c1 = t1 + D < n ? (unsigned char)T[t1 + D] : -1;
        const int c2 = t2 + D < n ? (unsigned char)T[t2 + D] : -1;
        at_node = c1 != c2;
      }
      s = at_node ? D - 1 : par - 1;
    }
  }
  /* filter (b), :349-375: position i+1 loses I1 when position i holds I with equal t and l >= */
  long total = 0;
  for (size_t i = m; i-- > 1;) {
    if (!cnt[i]) continue;
    if (i == 0) break;
    const occ_t* prev = lists[i - 1]; const size_t np = cnt[i - 1];
    size_t w = 0;
    for (size_t k = 0; k < cnt[i]; ++k) {
      int rim = 0;
      for (size_t q = 0; q < np && !rim; ++q)
        if (prev[q].t == lists[i][k].t && prev[q].l >= lists[i][k].l) rim = 1;
      if (!rim) lists[i][w++] = lists[i][k];
    }
    /* the reference filters list i+1 against the still unfiltered list i: keep the pre-(b)
     * contents of `lists[i]` available for position i+1 -- processed already (descending i) */
    cnt[i] = w;
  }
  for (size_t i = 0; i < m; ++i) {
    for (size_t k = 0; k < cnt[i]; ++k) {
      if (total < cap) { out[3 * total] = (int32_t)i; out[3 * total + 1] = lists[i][k].t; out[3 * total + 2] = lists[i][k].l; }
      ++total;
    }
    free(lists[i]);
  }
  free(lists); free(cnt);
  return total;
}
