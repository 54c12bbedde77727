/* Post-refinement of the factorizations of one EST: validity / duplicate removal, recovery of lost
 * prefixes and suffixes, removal of false small exons, search for new small exons, final cleaning.
 * Behaviour follows src/factorization-refinement.c of the reference (cited per function), quirks
 * included.  Dynamic programs are backend (GPU) calls. */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "estfact.h"

#define UB_VERY_SMALL_EXON 2
#define LB_SMALL_EXON 6
#define UB_SMALL_EXON 23
#define UB_MED_EXON 100
#define AFFIXES_LENGTH 5
#define MAX_ERROR_RATE 0.17
#define MIN_PERFECT_BORDER 6
#define MAX_ERRORS_AS_SMALL 2
#define INTRON_ND 2

static size_t zmin(size_t a, size_t b) { return a < b ? a : b; }
static size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

/* one question over the sequences themselves; 0 = answered, EF_DP_PENDING = noted (collect mode, estfact.h: ef_ahead) */
static int dp(ef_backend* be, int kind, const char* a, size_t la, const char* b, size_t lb,
              uint32_t p0, uint32_t p1, uint32_t p2, uint32_t tail, ef_dp_res* r) {
  ef_dp_req rq = { kind, a, la, b, lb, p0, p1, p2, tail, 0 };
  const int rc = ef_dp_one(be, &rq, r);
  if (rc != 0 && rc != EF_DP_PENDING) { fprintf(stderr, "* FATAL dynamic-programming backend failed (kind %d)\n", kind); abort(); }
  return rc;
}

/* compute_edit_distance (src/compute-alignments.c:240-249) as a request: false when the strings are
 * equal (distance 0 without a dynamic program) */
static bool ed_request(ef_dp_req* q, const char* a, size_t la, const char* b, size_t lb) {
  if (la == lb && strncmp(a, b, la) == 0) return false;
  const ef_dp_req r = { EF_DP_ED, a, la, b, lb, 0, 0, 0, 0, 0 };
  *q = r;
  return true;
}
static int dp_many(ef_backend* be, const ef_dp_req* q, ef_dp_res* r, size_t n) {
  const int rc = ef_dp_many(be, q, r, n);
  if (rc != 0 && rc != EF_DP_PENDING) { fprintf(stderr, "* FATAL dynamic-programming backend failed\n"); abort(); }
  return rc;
}

/* valid bytes after t[len] (0..2) when t points into the NUL-terminated string s */
static uint32_t tail_of(const char* s, const char* t, size_t len) {
  const size_t total = ef_genomic_len(s), end = (size_t)(t - s) + len;
  if (end >= total) return 0;
  return total - end >= 2 ? 2u : 1u;
}

/* remove_factorizations_with_very_small_exons (:85-112) */
void ef_remove_factorizations_with_very_small_exons(ef_list* facts) {
  ef_iter it = efl_begin(facts);
  while (efi_has_next(&it)) {
    ef_list* f = (ef_list*)efi_next(&it);
    bool small = false;
    ef_iter fi = efl_begin(f);
    while (!small && efi_has_next(&fi)) { const ef_factor* x = (const ef_factor*)efi_next(&fi); small = (x->EST_end + 1 - x->EST_start) <= UB_VERY_SMALL_EXON; }
    if (small) efi_remove(&it, ef_factorization_free);
  }
}

/* remove_invalid_factorizations (:121-165) */
static void remove_invalid(ef_list* facts) {
  ef_iter it = efl_begin(facts);
  while (efi_has_next(&it)) {
    ef_list* f = (ef_list*)efi_next(&it);
    bool invalid = false;
    const ef_factor* prev = NULL;
    ef_iter fi = efl_begin(f);
    while (!invalid && efi_has_next(&fi)) {
      const ef_factor* x = (const ef_factor*)efi_next(&fi);
      invalid = (x->EST_start > x->EST_end) || (x->GEN_start > x->GEN_end);
      if (!invalid && prev) invalid = (prev->EST_end >= x->EST_start) || (prev->GEN_end >= x->GEN_start);
      prev = x;
    }
    if (invalid) efi_remove(&it, ef_factorization_free);
  }
}

/* remove_duplicated_factorizations (:175-240): exact duplicates, the earlier one stays (the
 * reference's rotating-hash pre-check has no false negatives, so it never changes the result) */
void ef_remove_duplicated_factorizations(ef_list* facts) {
  ef_iter i1 = efl_begin(facts);
  while (efi_has_next(&i1)) {
    ef_list* f1 = (ef_list*)efi_next(&i1);
    ef_iter i2 = efl_begin(facts);
    while (efi_has_next(&i2)) {
      ef_list* f2 = (ef_list*)efi_next(&i2);
      if (f1 == f2) break;
      if (efl_size(f1) != efl_size(f2)) continue;
      bool equal = true;
      ef_iter a = efl_begin(f1), b = efl_begin(f2);
      while (equal && efi_has_next(&a)) {
        const ef_factor* x = (const ef_factor*)efi_next(&a); const ef_factor* y = (const ef_factor*)efi_next(&b);
        equal = x->EST_start == y->EST_start && x->EST_end == y->EST_end && x->GEN_start == y->GEN_start && x->GEN_end == y->GEN_end;
      }
      if (equal) { efi_remove(&i1, ef_factorization_free); break; }
    }
  }
}

/* recover_lost_prefixes_and_suffixes (:1177-1266) */
static void recover_affixes(const ef_seq* gen, ef_est* e, ef_backend* be) {
  const char* G = gen->seq;
  const char* E = e->info->seq;
  const size_t totg = ef_genomic_len(G), tote = strlen(E);
  ef_iter it = efl_begin(e->factorizations);
  while (efi_has_next(&it)) {
    ef_list* f = (ef_list*)efi_next(&it);
    /* the lost prefix and the lost suffix do not depend on each other: one request pair */
    ef_dp_req q[2]; ef_dp_res r[2]; size_t nq = 0;
    int sp = -1, ss = -1;
    char *ef = NULL, *gf = NULL;
    ef_factor* ff = (ef_factor*)efl_head(f);
    if (ff->EST_start > 0 && ff->GEN_start > 0) {
      const size_t flen = (size_t)(ff->EST_start < ff->GEN_start ? ff->EST_start : ff->GEN_start);
      const int cap = (int)((1.0 + MAX_ERROR_RATE) * flen);
      const size_t elen = (size_t)(ff->EST_start < cap ? ff->EST_start : cap);
      const size_t glen = (size_t)(ff->GEN_start < cap ? ff->GEN_start : cap);
      ef = (char*)malloc(elen + 1); gf = (char*)malloc(glen + 1);
      for (size_t i = 0; i < elen; ++i) ef[i] = E[ff->EST_start - 1 - i];
      for (size_t i = 0; i < glen; ++i) gf[i] = G[ff->GEN_start - 1 - i];
      ef[elen] = gf[glen] = '\0';
      if (ef[0] != gf[0]) {
        const ef_dp_req x = { EF_DP_AFFIX, ef, elen, gf, glen, 0, 0, 0, 0, 1 };     /* reversed copies */
        sp = (int)nq; q[nq++] = x;
      }
    }
    ef_factor* fl = (ef_factor*)efl_tail(f);
    if ((tote - (size_t)fl->EST_end) > 1 && (totg - (size_t)fl->GEN_end) > 1) {
      const size_t flen = zmin(tote - fl->EST_end - 1, totg - fl->GEN_end - 1);
      /* `(int)(1.0+_MAX_ERROR_RATE_)*flen` in the reference: the cast binds first => 1*flen */
      const size_t elen = zmin(tote - fl->EST_end - 1, (size_t)((int)(1.0 + MAX_ERROR_RATE)) * flen);
      const size_t glen = zmin(totg - fl->GEN_end - 1, (size_t)((int)(1.0 + MAX_ERROR_RATE)) * flen);
      const char* es = E + fl->EST_end;          /* starts ON the last exon character (:1239,1242) */
      const char* gs = G + fl->GEN_end;
      if (es[0] != gs[0]) {
        const ef_dp_req x = { EF_DP_AFFIX, es, elen, gs, glen, 0, 0, 0, 0, 0 };
        ss = (int)nq; q[nq++] = x;
      }
    }
    dp_many(be, q, r, nq);
    if (sp >= 0 && r[sp].v[0]) { ff->EST_start -= r[sp].v[1]; ff->GEN_start -= r[sp].v[2]; }
    if (ss >= 0 && r[ss].v[0]) { fl->EST_end += r[ss].v[1]; fl->GEN_end += r[ss].v[2]; }
    free(ef); free(gf);
  }
}

/* analyze_possibly_small_exon (:960-1093); `it` stands right after `next` */
static bool analyze_small_exon(ef_factor** pprev, ef_factor** pcurr, ef_factor* next, ef_iter* it,
                               const ef_seq* gen, ef_est* e, ef_backend* be) {
  ef_factor* prev = *pprev; ef_factor* curr = *pcurr;
  if (prev == NULL || next == NULL) return false;
  const char* G = gen->seq; const char* E = e->info->seq;
  const size_t elen = (size_t)(curr->EST_end + 1 - curr->EST_start);
  const size_t glen = (size_t)(curr->GEN_end + 1 - curr->GEN_start);
  if (elen > UB_MED_EXON) return false;
  const int estart_i = prev->EST_start + 1 > prev->EST_end + 1 - AFFIXES_LENGTH ? prev->EST_start + 1 : prev->EST_end + 1 - AFFIXES_LENGTH;
  const size_t estart = (size_t)estart_i;
  const size_t eend = zmin((size_t)next->EST_end, (size_t)(next->EST_start + AFFIXES_LENGTH));
  const size_t epreflen = prev->EST_end + 1 - estart, esufflen = eend - next->EST_start, allelen = eend - estart;
  const char* allefact = E + estart;
  const int gstart_i = prev->GEN_start + 1 > prev->GEN_end + 1 - AFFIXES_LENGTH ? prev->GEN_start + 1 : prev->GEN_end + 1 - AFFIXES_LENGTH;
  const size_t gstart = (size_t)gstart_i;
  const size_t gend = zmin((size_t)next->GEN_end, (size_t)(next->GEN_start + AFFIXES_LENGTH));
  const size_t gpreflen = prev->GEN_end + 1 - gstart, gsufflen = gend - next->GEN_start, allglen = gend - gstart;
  const char* allgfact = G + gstart;
  /* three independent edit distances (:990,1012,1017), requested together */
  ef_dp_req q3[3]; ef_dp_res r3[3]; size_t n3 = 0;
  int s_orig = -1, s_pref = -1, s_suff = -1;
  if (ed_request(&q3[n3], E + curr->EST_start, elen, G + curr->GEN_start, glen)) s_orig = (int)n3++;
  if (ed_request(&q3[n3], allefact, epreflen, allgfact, gpreflen)) s_pref = (int)n3++;
  /* the reference takes the "suffix" BEFORE the window start (allefact - esufflen, :1017-1018) */
  if (estart >= esufflen && gstart >= gsufflen &&
      ed_request(&q3[n3], allefact - esufflen, esufflen, allgfact - gsufflen, gsufflen)) s_suff = (int)n3++;
  if (dp_many(be, q3, r3, n3) == EF_DP_PENDING) return false;
  const size_t orig_ed = s_orig >= 0 ? (size_t)(uint32_t)r3[s_orig].v[0] : 0;
  const size_t ed_pref = s_pref >= 0 ? (size_t)(uint32_t)r3[s_pref].v[0] : 0;
  const size_t ed_suff = s_suff >= 0 ? (size_t)(uint32_t)r3[s_suff].v[0] : 0;
  ef_dp_res r;
  const uint32_t max_errs = (uint32_t)(orig_ed + ed_pref + ed_suff);
  if (dp(be, EF_DP_BORDERS, allefact, allelen, allgfact, allglen, 0, (uint32_t)allelen, max_errs, tail_of(G, allgfact, allglen), &r) == EF_DP_PENDING) return false;
  if (ef_collecting(be)) return false;                  /* every answer is known: nothing is changed in this mode */
  if (!r.v[0]) return false;
  const size_t off_p = (size_t)r.v[1], off_t1 = (size_t)r.v[2], off_t2 = (size_t)r.v[3];
  const double prev_avg = (ef_burset_adaptor(G, (size_t)(prev->GEN_end + 1), (size_t)curr->GEN_start) +
                           ef_burset_adaptor(G, (size_t)(curr->GEN_end + 1), (size_t)next->GEN_start)) / 2.0;
  const double new_freq = ef_burset_adaptor(G, gstart + off_t1, gend - allglen + off_t2);
  if (!(new_freq >= prev_avg)) return false;
  prev->EST_end = (int)(estart + off_p - 1);
  next->EST_start = (int)(eend + off_p - allelen);
  prev->GEN_end = (int)(gstart + off_t1 - 1);
  next->GEN_start = (int)(gend + off_t2 - allglen);
  efi_prev(it);
  efi_remove(it, free);                      /* drops `curr` */
  *pcurr = prev;
  efi_prev(it);
  *pprev = (ef_factor*)it->prev->el;         /* NULL at the list head (sentinel) */
  efi_next(it);
  efi_next(it);
  return true;
}

/* remove_false_small_exons (:1095-1125) */
static void remove_false_small_exons(const ef_seq* gen, ef_est* e, ef_backend* be) {
  ef_iter it = efl_begin(e->factorizations);
  while (efi_has_next(&it)) {
    ef_list* f = (ef_list*)efi_next(&it);
    ef_factor *prev = NULL, *curr = NULL, *next = NULL;
    bool removed = false;
    ef_iter fi = efl_begin(f);
    if (efi_has_next(&fi)) next = (ef_factor*)efi_next(&fi);
    while (next != NULL) {
      if (!removed) {
        prev = curr; curr = next; next = NULL;
        if (efi_has_next(&fi)) next = (ef_factor*)efi_next(&fi);
      }
      removed = analyze_small_exon(&prev, &curr, next, &fi, gen, e, be);
    }
  }
}

static int base2(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
static int kmer_code(const char* s) {
  int code = 0;
  for (int k = 0; k < LB_SMALL_EXON; ++k) { const int b = base2(s[k]); if (b < 0) return -1; code = (code << 2) | b; }
  return code;
}
/* first index in the ascending array a[0..n) whose value is >= v */
static size_t lower_bound_u32(const uint32_t* a, size_t n, uint32_t v) {
  size_t lo = 0, hi = n;
  while (lo < hi) { const size_t mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}

static bool canonical_intron(const char* G, size_t s, size_t e) {           /* :481-494 */
  return (G[s] == 'G' && G[s + 1] == 'T' && G[e - 1] == 'A' && G[e] == 'G') ||
         (G[s] == 'g' && G[s + 1] == 't' && G[e - 1] == 'a' && G[e] == 'g');
}

static ef_factor* factor_new(int es, int ee, int gs, int ge) {
  ef_factor* f = (ef_factor*)malloc(sizeof(ef_factor));
  f->EST_start = es; f->EST_end = ee; f->GEN_start = gs; f->GEN_end = ge;
  return f;
}

/* search_small_exon_at_prefix (:500-606) */
static void small_exon_at_prefix(ef_factor* p1, ef_iter* it, const ef_seq* gen, ef_est* e,
                                 const ef_config* cfg, ef_backend* be) {
  const char* G = gen->seq; const char* E = e->info->seq;
  const size_t e1len = (size_t)(p1->EST_end + 1 - p1->EST_start), g1len = (size_t)(p1->GEN_end + 1 - p1->GEN_start);
  if (!((e1len + (size_t)p1->EST_start) >= (LB_SMALL_EXON + UB_SMALL_EXON))) return;
  const size_t eplen = zmin(zmin((size_t)p1->EST_start, (size_t)p1->GEN_start), 2 * UB_SMALL_EXON);
  const char* epfact = E + p1->EST_start - eplen;
  const size_t e1plen = zmin(zmin(e1len, g1len), UB_SMALL_EXON);
  /* the common factor and the edit distance of the exon prefix (needed only when the factor is long
   * enough, :546) do not depend on each other: requested together */
  ef_dp_req q2[2]; ef_dp_res r2[2]; size_t n2 = 1;
  { const ef_dp_req x = { EF_DP_LCF, G, (size_t)p1->GEN_start, epfact, eplen, 0, 0, 0, 0, 0 }; q2[0] = x; }
  const bool need_ed = ed_request(&q2[1], E + p1->EST_start, e1plen, G + p1->GEN_start, e1plen);
  if (need_ed) n2 = 2;
  if (dp_many(be, q2, r2, n2) == EF_DP_PENDING) return;
  ef_dp_res r = r2[0];
  const size_t cflen = (size_t)r.v[0], pg = (size_t)r.v[1], pe = (size_t)r.v[2];
  if (cflen < LB_SMALL_EXON) return;
  const unsigned edp = need_ed ? (unsigned)r2[1].v[0] : 0u;
  /* `pe` is an offset inside the discarded prefix but the reference uses it as an absolute EST
   * coordinate from here on (:551-603); reproduced as is */
  const size_t allelen = zmin((size_t)(p1->EST_end + 1), (size_t)(p1->EST_start + UB_SMALL_EXON)) - pe;
  const size_t allglen = zmin((size_t)(p1->GEN_end + 1), (size_t)(p1->GEN_start + UB_SMALL_EXON)) - pg;
  if (allelen < 2 * LB_SMALL_EXON || allelen > 4096) return;     /* outside what the reference can evaluate */
  if (dp(be, EF_DP_BORDERS, E + pe, allelen, G + pg, allglen, LB_SMALL_EXON, (uint32_t)(allelen - LB_SMALL_EXON), edp,
         tail_of(G, G + pg, allglen), &r) == EF_DP_PENDING) return;
  if (ef_collecting(be)) return;
  if (!r.v[0]) return;
  const size_t off_p = (size_t)r.v[1], off_t1 = (size_t)r.v[2], off_t2 = (size_t)r.v[3];
  if ((int)off_t2 - (int)off_t1 < cfg->min_intron_length) return;
  if (!canonical_intron(G, pg + off_t1, pg + off_t2 - 1)) return;
  if (off_p - pe < LB_SMALL_EXON) return;                        /* size_t arithmetic as in the reference */
  ef_factor* nw = factor_new((int)pe, (int)(pe + off_p - 1), (int)pg, (int)(pg + off_t1 - 1));
  p1->EST_start = (int)(pe + off_p);
  p1->GEN_start = (int)(pg + off_t2);
  efi_insert_before(it, nw);
}

/* search_small_exon (:641-873) */
static void small_exon_between(ef_factor* p1, ef_factor* p2, ef_iter* it, const ef_seq* gen, ef_est* e,
                               const ef_config* cfg, ef_backend* be) {
  const char* G = gen->seq; const char* E = e->info->seq;
  const size_t e1len = (size_t)(p1->EST_end + 1 - p1->EST_start), g1len = (size_t)(p1->GEN_end + 1 - p1->GEN_start);
  const size_t e2len = (size_t)(p2->EST_end + 1 - p2->EST_start), g2len = (size_t)(p2->GEN_end + 1 - p2->GEN_start);
  if (!((e1len + e2len) >= (LB_SMALL_EXON + 2 * UB_SMALL_EXON))) return;
  const size_t e1slen = zmin(zmin(e1len, g1len), UB_SMALL_EXON), g1slen = e1slen;
  const size_t e1sstart = (size_t)p1->EST_end + 1 - e1slen, g1sstart = (size_t)p1->GEN_end + 1 - g1slen;
  /* the border strings lie inside the two exons: they are read in place (the reference copies them) */
  const char* e1s = E + e1sstart;
  const char* g1s = G + g1sstart;
  const size_t e2plen = zmin(zmin(e2len, g2len), UB_SMALL_EXON), g2plen = e2plen;
  const size_t e2pstart = (size_t)p2->EST_start, g2pstart = (size_t)p2->GEN_start;
  const char* e2p = E + e2pstart;
  const char* g2p = G + g2pstart;
  /* both edit distances and, for the side(s) that are not identical, the common factors the
   * reference asks for afterwards (:690-718): independent of each other, requested together */
  ef_dp_req q4[4]; ef_dp_res r4[4]; size_t n4 = 0;
  int s_sed = -1, s_ped = -1, s_l1 = -1, s_l2 = -1;
  if (ed_request(&q4[n4], e1s, e1slen, g1s, g1slen)) s_sed = (int)n4++;
  if (ed_request(&q4[n4], e2p, e2plen, g2p, g2plen)) s_ped = (int)n4++;
  if (s_sed >= 0) { const ef_dp_req x = { EF_DP_LCF, e1s, e1slen, g1s, g1slen, 0, 0, 0, 0, 0 }; s_l1 = (int)n4; q4[n4++] = x; }
  if (s_ped >= 0) { const ef_dp_req x = { EF_DP_LCF, e2p, e2plen, g2p, g2plen, 0, 0, 0, 0, 0 }; s_l2 = (int)n4; q4[n4++] = x; }
  if (dp_many(be, q4, r4, n4) == EF_DP_PENDING) return;
  if (ef_collecting(be)) return;                       /* (no further question follows) */
  const size_t sed = s_sed >= 0 ? (size_t)(uint32_t)r4[s_sed].v[0] : 0;
  const size_t ped = s_ped >= 0 ? (size_t)(uint32_t)r4[s_ped].v[0] : 0;
  bool go = false;
  ef_phase(EFP_NS_BETWEEN_CLASS);
  const int orig_class = ef_classify_intron(gen, p1->GEN_end + 1, p2->GEN_start - 1);
  if (sed + ped > MAX_ERRORS_AS_SMALL) go = true;
  if (orig_class == INTRON_ND) go = true;
  if (go) {
    ef_phase(EFP_NS_BETWEEN_SEARCH);
    size_t e1socc = 0, g1socc = 0, f1slen = e1slen;
    if (sed > 0) { const ef_dp_res r = r4[s_l1]; f1slen = (size_t)r.v[0]; e1socc = (size_t)r.v[1]; g1socc = (size_t)r.v[2]; }
    size_t e2pocc = 0, g2pocc = 0, f2plen = e2plen;
    if (ped > 0) { const ef_dp_res r = r4[s_l2]; f2plen = (size_t)r.v[0]; e2pocc = (size_t)r.v[1]; g2pocc = (size_t)r.v[2]; }
    if (f1slen == e1slen && e2pocc > 0) {
      size_t nf = f1slen + 1;
      while ((nf - f1slen) < e2pocc && E[e1sstart + e1socc + f1slen] == G[g2pstart + nf - f1slen]) ++nf;
      if (nf - 1 > f1slen) f1slen = nf - 1;
    }
    const size_t elen = (e1slen - e1socc) + (e2pocc + f2plen) - (2 * MIN_PERFECT_BORDER);
    const size_t estart = e1sstart + e1socc + MIN_PERFECT_BORDER;
    const size_t allgstart = g1sstart + g1socc + MIN_PERFECT_BORDER;
    const size_t allglen = g2pstart + g2pocc + f2plen - MIN_PERFECT_BORDER - allgstart;
    const size_t MIL = zmax(4, (size_t)cfg->min_intron_length);
    if (f1slen < MIN_PERFECT_BORDER) go = false;
    else if (f2plen < MIN_PERFECT_BORDER) go = false;
    else if (allglen < 2 * MIL + LB_SMALL_EXON) go = false;
    else if (elen < LB_SMALL_EXON) go = false;
    if (go) {
      char* efact = ef_real_substring((int)estart, (int)elen, E);
      /* the intron is read where it lies; the copy the reference makes (real_substring of up to 20 kb) is only
       * needed by the strstr() fallback, which wants to cut it with a terminator */
      const bool in_place = allgstart + allglen <= ef_genomic_len(G);
      char* allg_copy = NULL;
      const char* allg = in_place ? G + allgstart : (allg_copy = ef_real_substring((int)allgstart, (int)allglen, G));
      size_t max_len = 0, ecut1 = 0, ecut2 = 0, gcut1_1 = 0, gcut1_2 = 0, gcut2_1 = 0, gcut2_2 = 0;
      const size_t max_offstart = zmin(zmin(f1slen + 1 - MIN_PERFECT_BORDER, elen + 1 - LB_SMALL_EXON), allglen + 1 - (2 * MIL) - LB_SMALL_EXON);
      /* The reference runs strstr() over the whole intron for every (offstart, offend) pair
       * (:781-834).  Same occurrences, same order, found through the 6-mer index of the genomic
       * sequence (gen->kmer_*, built once per gene; every pattern is at least LB_SMALL_EXON = 6
       * long): the candidates of a window are a contiguous run of the 6-mer's ascending position
       * list, and how far a candidate matches the pattern of this offstart is computed once for all
       * offends (a pattern of length plen occurs there iff the match is at least plen long).
       * Patterns whose first six characters are not all ACGT fall back to strstr().
       * A pair whose pattern is not longer than the best found so far cannot replace it (strict
       * comparison at :820): such pairs -- all later offends of this offstart, and every later offstart
       * once elen - os itself is not longer -- are skipped. */
      /* The index lookups of all offstarts are independent of each other and every one of them is a chain of
       * cache misses (offset table, position list, the characters behind each occurrence): they are issued for
       * all offstarts first, stage by stage with prefetches, and the decision loops below find the lists ready. */
      ef_phase(EFP_TMP1);
      enum { OS_MAX = 24, CAND_MAX = 12 };
      int os_code[OS_MAX]; const uint32_t* os_all[OS_MAX]; uint32_t os_nall[OS_MAX];
      uint32_t os_q[OS_MAX][CAND_MAX], os_m[OS_MAX][CAND_MAX]; unsigned char os_nc[OS_MAX]; bool os_listed[OS_MAX];
      const size_t n_os = max_offstart < OS_MAX ? max_offstart : OS_MAX;
      for (size_t os = 0; os < n_os; ++os) {
        os_code[os] = kmer_code(efact + os);
        if (os_code[os] >= 0) __builtin_prefetch(&gen->kmer_first[os_code[os]], 0, 3);
      }
      for (size_t os = 0; os < n_os; ++os) {
        if (os_code[os] < 0) continue;
        os_all[os] = gen->kmer_pos + gen->kmer_first[os_code[os]];
        os_nall[os] = gen->kmer_first[os_code[os] + 1] - gen->kmer_first[os_code[os]];
        for (uint32_t b = 0; b < os_nall[os]; b += 16) __builtin_prefetch(os_all[os] + b, 0, 3);
        if (os_nall[os]) __builtin_prefetch(os_all[os] + os_nall[os] - 1, 0, 3);
      }
      for (size_t os = 0; os < n_os; ++os) {
        os_nc[os] = 0; os_listed[os] = false;
        if (os_code[os] < 0) continue;
        const uint32_t* all = os_all[os];
        size_t k = lower_bound_u32(all, os_nall[os], (uint32_t)(allgstart + os + MIL));     /* text_lo of every offend */
        os_listed[os] = true;
        for (; k < os_nall[os]; ++k) {
          const size_t qa = all[k];
          if (qa >= allgstart + allglen) break;
          if (os_nc[os] == CAND_MAX) { os_listed[os] = false; break; }
          os_q[os][os_nc[os]++] = (uint32_t)(qa - allgstart);
          __builtin_prefetch(allg + (qa - allgstart) + LB_SMALL_EXON, 0, 3);
        }
      }
      for (size_t os = 0; os < n_os; ++os) {
        if (!os_listed[os]) continue;
        const size_t pmax = elen - os;                       /* the longest pattern of this offstart */
        for (size_t c = 0; c < os_nc[os]; ++c) {
          const size_t q = os_q[os][c];
          size_t room = allglen - q; if (room > pmax) room = pmax;
          size_t m = LB_SMALL_EXON;
          while (m < room && allg[q + m] == efact[os + m]) ++m;
          os_m[os][c] = (uint32_t)m;
        }
      }
      ef_phase(EFP_TMP2);
      for (size_t os = 0; os < max_offstart && elen - os > max_len; ++os) {
        const size_t max_offend = zmin(zmin(f2plen + 1 - MIN_PERFECT_BORDER, elen + 1 - os - LB_SMALL_EXON), allglen + 1 - (2 * MIL) - LB_SMALL_EXON - os);
        const int code = os < n_os ? os_code[os] : kmer_code(efact + os);
        const bool listed = os < n_os && os_listed[os];
        const uint32_t* cq = os < n_os ? os_q[os] : NULL; const uint32_t* cm = os < n_os ? os_m[os] : NULL;
        const size_t nc = listed ? os_nc[os] : 0;
        if (code >= 0 && !listed) {                            /* more occurrences than the table holds: as before, one by one */
          const uint32_t* all = gen->kmer_pos + gen->kmer_first[code];
          const size_t nall = gen->kmer_first[code + 1] - gen->kmer_first[code];
          const size_t from = lower_bound_u32(all, nall, (uint32_t)allgstart);
          const uint32_t* cand = all + from; const size_t ncand = nall - from;
          for (size_t oe = 0; oe < max_offend && elen - os - oe > max_len; ++oe) {
            const size_t plen = elen - os - oe;
            const size_t text_lo = os + MIL, text_hi = allglen - oe - MIL;
            for (size_t cursor = 0; cursor < ncand; ++cursor) {
              const size_t qa = cand[cursor];
              if (qa >= allgstart + allglen) break;
              const size_t q = qa - allgstart;
              if (q < text_lo || plen > text_hi || q > text_hi - plen) continue;
              if (memcmp(allg + q + LB_SMALL_EXON, efact + os + LB_SMALL_EXON, plen - LB_SMALL_EXON) != 0) continue;
              const size_t i1s = allgstart + os, i1e = allgstart + q - 1;
              const size_t i2s = i1e + 1 + elen - os - oe, i2e = allgstart + allglen - oe - 1;
              const int t1 = ef_classify_intron(gen, (int)i1s, (int)i1e), t2 = ef_classify_intron(gen, (int)i2s, (int)i2e);
              if (t1 != INTRON_ND && t2 != INTRON_ND && plen > max_len) {
                max_len = plen; ecut1 = estart + os; ecut2 = estart + os + plen; gcut1_1 = i1s; gcut1_2 = i1e + 1; gcut2_1 = i2s; gcut2_2 = i2e + 1;
              }
            }
          }
          continue;
        }
        /* no listed candidate matches further than `reach`: the offends whose pattern is longer find nothing */
        size_t oe0 = 0;
        if (code >= 0) {
          size_t reach = 0;
          for (size_t c = 0; c < nc; ++c) if (cm[c] > reach) reach = cm[c];
          if (reach < LB_SMALL_EXON) continue;                 /* no occurrence at all */
          if (elen - os > reach) oe0 = elen - os - reach;
        }
        for (size_t oe = oe0; oe < max_offend && elen - os - oe > max_len; ++oe) {
          const size_t plen = elen - os - oe;                 /* pattern efact[os .. elen-oe) */
          const size_t text_lo = os + MIL, text_hi = allglen - oe - MIL;   /* text allg[text_lo .. text_hi) */
          if (code >= 0) {
            if (plen > text_hi) continue;
            for (size_t c = 0; c < nc; ++c) {
              const size_t q = cq[c];
              if (q > text_hi - plen) break;                   /* ascending: the rest lies further right */
              if (cm[c] < plen) continue;                      /* (q >= text_lo by construction of the list) */
              const size_t i1s = allgstart + os, i1e = allgstart + q - 1;
              const size_t i2s = i1e + 1 + elen - os - oe, i2e = allgstart + allglen - oe - 1;
              const int t1 = ef_classify_intron(gen, (int)i1s, (int)i1e), t2 = ef_classify_intron(gen, (int)i2s, (int)i2e);
              if (t1 != INTRON_ND && t2 != INTRON_ND && plen > max_len) {
                max_len = plen; ecut1 = estart + os; ecut2 = estart + os + plen; gcut1_1 = i1s; gcut1_2 = i1e + 1; gcut2_1 = i2s; gcut2_2 = i2e + 1;
              }
            }
            continue;
          }
          /* The first six characters of the pattern hold a byte that is no upper-case A, C, G or T (an N of the
           * EST, mostly): strstr() compares bytes, so the pattern can only occur where the genomic sequence has
           * that very byte -- its positions are listed per byte value (gen->other_*), nearly always none.  (The
           * reference's strstr() over the whole intron for each of these pairs was 4 us a call and most of this
           * function's time.) */
          if (plen > text_hi) continue;
          size_t j = 0;
          while (base2(efact[os + j]) >= 0) ++j;               /* < LB_SMALL_EXON: kmer_code said so */
          if (!in_place || !gen->other_first) {                /* a window cut by the end of the sequence: as the reference does it */
            if (!allg_copy) allg_copy = ef_real_substring((int)allgstart, (int)allglen, G);
            const char sv_e = efact[elen - oe], sv_g = allg_copy[text_hi];
            efact[elen - oe] = '\0'; allg_copy[text_hi] = '\0';
            for (char* occ = allg_copy + text_lo; (occ = strstr(occ, efact + os)) != NULL; ++occ) {
              const size_t q = (size_t)(occ - allg_copy);
              const size_t i1s = allgstart + os, i1e = allgstart + q - 1;
              const size_t i2s = i1e + 1 + elen - os - oe, i2e = allgstart + allglen - oe - 1;
              const int t1 = ef_classify_intron(gen, (int)i1s, (int)i1e), t2 = ef_classify_intron(gen, (int)i2s, (int)i2e);
              if (t1 != INTRON_ND && t2 != INTRON_ND && plen > max_len) {
                max_len = plen; ecut1 = estart + os; ecut2 = estart + os + plen; gcut1_1 = i1s; gcut1_2 = i1e + 1; gcut2_1 = i2s; gcut2_2 = i2e + 1;
              }
            }
            efact[elen - oe] = sv_e; allg_copy[text_hi] = sv_g;
            continue;
          }
          const unsigned char bad = (unsigned char)efact[os + j];
          if (bad == 0) continue;                              /* (a pattern cut by the end of the EST: strstr would look for less; not reached: elen is inside the EST) */
          const uint32_t* ol = gen->other_pos + gen->other_first[bad];
          const size_t nol = gen->other_first[bad + 1] - gen->other_first[bad];
          for (size_t k = lower_bound_u32(ol, nol, (uint32_t)(allgstart + text_lo + j)); k < nol; ++k) {
            const size_t q = ol[k] - j - allgstart;             /* window-relative start of the would-be occurrence */
            if (q > text_hi - plen) break;
            if (memcmp(allg + q, efact + os, plen) != 0) continue;
            const size_t i1s = allgstart + os, i1e = allgstart + q - 1;
            const size_t i2s = i1e + 1 + elen - os - oe, i2e = allgstart + allglen - oe - 1;
            const int t1 = ef_classify_intron(gen, (int)i1s, (int)i1e), t2 = ef_classify_intron(gen, (int)i2s, (int)i2e);
            if (t1 != INTRON_ND && t2 != INTRON_ND && plen > max_len) {
              max_len = plen; ecut1 = estart + os; ecut2 = estart + os + plen; gcut1_1 = i1s; gcut1_2 = i1e + 1; gcut2_1 = i2s; gcut2_2 = i2e + 1;
            }
          }
        }
      }
      ef_phase(EFP_TMP3);
      if (max_len >= LB_SMALL_EXON) {
        ef_factor* nw = factor_new((int)ecut1, (int)ecut2 - 1, (int)gcut1_2, (int)gcut2_1 - 1);
        p2->EST_start = (int)ecut2; p2->GEN_start = (int)gcut2_2;
        p1->EST_end = (int)ecut1 - 1; p1->GEN_end = (int)gcut1_1 - 1;
        efi_insert_before(it, nw);
      }
      free(efact); free(allg_copy);
    }
  }
}

/* search_for_new_small_exons (:875-912) */
static void search_new_small_exons(const ef_seq* gen, ef_est* e, const ef_config* cfg, ef_backend* be) {
  ef_iter it = efl_begin(e->factorizations);
  while (efi_has_next(&it)) {
    ef_list* f = (ef_list*)efi_next(&it);
    ef_factor *p1 = NULL, *p2 = NULL;
    ef_iter fi = efl_begin(f);
    if (efi_has_next(&fi)) {
      p1 = (ef_factor*)efi_next(&fi);
      if (p1->EST_start > LB_SMALL_EXON) { ef_phase(EFP_NS_PREFIX); small_exon_at_prefix(p1, &fi, gen, e, cfg, be); ef_phase(EFP_REF_NEW_SMALL); }
    }
    if (efi_has_next(&fi)) p2 = (ef_factor*)efi_next(&fi);
    while (p2 != NULL) {
      ef_phase(EFP_NS_BETWEEN_ASK);
      small_exon_between(p1, p2, &fi, gen, e, cfg, be);
      ef_phase(EFP_REF_NEW_SMALL);
      p1 = p2; p2 = NULL;
      if (efi_has_next(&fi)) p2 = (ef_factor*)efi_next(&fi);
    }
  }
}

/* clean_factorizations (:914-950): cleaning on the ORIGINAL EST sequence */
static void clean_factorizations(const ef_seq* gen, ef_est* e, const ef_config* cfg, ef_backend* be) {
  ef_list* cleaned = efl_new();
  ef_iter it = efl_begin(e->factorizations);
  while (efi_has_next(&it)) {
    ef_list* f = (ef_list*)efi_next(&it);
    f = ef_clean_noisy_exons(f, gen->seq, e->info->original_seq, false, be);
    f = ef_clean_external_exons(f, gen->seq, e->info->original_seq, be);
    if (efl_empty(f)) efi_remove(&it, ef_factorization_free);
    else {
      bool added;
      cleaned = ef_add_if_not_exists(f, cleaned, cfg, &added);
      if (!added) efi_remove(&it, ef_factorization_free);
    }
  }
  efl_free(e->factorizations, NULL);
  e->factorizations = cleaned;
}

/* refine_EST_factorizations (:1270-1305) */
void ef_refine_est_factorizations(const ef_seq* gen, ef_est* e, const ef_config* cfg, ef_backend* be) {
  remove_invalid(e->factorizations);
  ef_remove_duplicated_factorizations(e->factorizations);
  ef_phase(EFP_REF_AFFIX);
  recover_affixes(gen, e, be);
  if (be->ahead) {
    /* what the two small-exon passes below are going to ask first (edit distances, common factors), for all exons
     * and introns of all factorizations in one request; then, with those answers, their border refinements in a
     * second one.  The passes themselves follow and find the answers (a pass that changes an exon makes the
     * questions of its neighbours different ones: those are asked when their turn comes, as before). */
    ef_phase(EFP_REF_FALSE_SMALL);
    for (int round = 0; round < 2; ++round) {
      ef_ahead_collect(be, true);
      remove_false_small_exons(gen, e, be);
      search_new_small_exons(gen, e, cfg, be);
      ef_ahead_collect(be, false);
      if (be->ahead->n_pending == 0) break;
      if (ef_ahead_flush(be) != 0) { fprintf(stderr, "* FATAL dynamic-programming backend failed\n"); abort(); }
    }
  }
  ef_phase(EFP_REF_FALSE_SMALL);
  remove_false_small_exons(gen, e, be);
  ef_remove_duplicated_factorizations(e->factorizations);
  ef_phase(EFP_REF_NEW_SMALL);
  search_new_small_exons(gen, e, cfg, be);
  ef_phase(EFP_REF_CLEAN);
  clean_factorizations(gen, e, cfg, be);
  ef_phase(EFP_FACTREF);
}
