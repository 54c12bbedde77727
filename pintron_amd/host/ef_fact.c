/* From the MEG to the factorizations of one EST: embedding enumeration, candidate factorizations,
 * cleaning steps, selection filters, intron refinement, polyA detection.
 * Behaviour follows src/est-factorizations.c, src/exon-complexity.c, src/detect-polya.c and
 * src/list.c:306-483 of the reference (cited per function); data structures are our own.  Every
 * dynamic program goes through the backend (ef_backend.dp), never computed here. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "estfact.h"

typedef struct { int p, t, l; } ptl;     /* pairing inside an embedding */

/* ---------------------------------------------------------------------------------------------- */
/* small helpers                                                                                  */
/* ---------------------------------------------------------------------------------------------- */
/* strlen(real_substring(index, length, s)) for index >= 0 without the copy: the substring read in
 * place ends at the terminator like the copy would */
static size_t view_len(const char* s, int index, int length) {
  if (length <= 0) return 0;
  const char* z = (const char*)memchr(s + index, 0, (size_t)length);     /* vectorised: this runs twice per exon */
  return z ? (size_t)(z - (s + index)) : (size_t)length;
}

char* ef_real_substring(int index, int length, const char* s) {     /* src/util.c:138-158 */
  if (index < 0) { length += index; index = 0; }
  if (length < 0) length = 0;
  char* r = (char*)malloc((size_t)length + 1);
  strncpy(r, s + index, (size_t)length);
  r[length] = '\0';
  return r;
}

static ef_factor* factor_new(int es, int ee, int gs, int ge) {
  ef_factor* f = (ef_factor*)malloc(sizeof(ef_factor));
  f->EST_start = es; f->EST_end = ee; f->GEN_start = gs; f->GEN_end = ge;
  return f;
}

void ef_factorization_free(void* fact) { efl_free((ef_list*)fact, free); }

void ef_est_free(ef_est* e) {
  if (!e) return;
  efl_free(e->factorizations, ef_factorization_free);
  efl_free(e->polyA_signals, NULL);
  efl_free(e->polyadenil_signals, NULL);
  free(e);
}

/* ---- questions asked ahead (estfact.h: ef_ahead) ------------------------------------------------ */
static void ahead_note(ef_ahead* ah, const ef_dp_req* q) {
  for (int k = 0; k < ah->n_pending; ++k) if (ef_req_same(&ah->q[ah->n + k], q)) return;
  if (ah->n + ah->n_pending < EF_AHEAD_MAX) ah->q[ah->n + ah->n_pending++] = *q;
}

int ef_dp_many_ahead(ef_backend* be, const ef_dp_req* reqs, ef_dp_res* res, size_t n) {
  ef_ahead* ah = be->ahead;
  enum { CAP = 32 };
  if (n > CAP) return ah->collecting ? EF_DP_PENDING : ef_backend_ask(be, reqs, res, n);
  size_t miss[CAP], nm = 0;
  for (size_t k = 0; k < n; ++k) {
    const int32_t* v = ef_ahead_find(ah, &reqs[k]);
    if (v) memcpy(res[k].v, v, sizeof res[k].v);
    else miss[nm++] = k;
  }
  if (ef_prof_on) { ef_prof.ahead_hits += n - nm; ef_prof.ahead_misses += ah->collecting ? 0 : nm; }
  if (nm == 0) return 0;
  if (ah->collecting) {
    for (size_t k = 0; k < nm; ++k) if (ef_req_keepable(&reqs[miss[k]])) ahead_note(ah, &reqs[miss[k]]);
    return EF_DP_PENDING;
  }
  if (nm == n) return ef_backend_ask(be, reqs, res, n);
  ef_dp_req mq[CAP]; ef_dp_res mr[CAP];
  for (size_t k = 0; k < nm; ++k) mq[k] = reqs[miss[k]];
  memset(mr, 0, nm * sizeof(ef_dp_res));
  const int rc = ef_backend_ask(be, mq, mr, nm);
  if (rc != 0) return rc;
  for (size_t k = 0; k < nm; ++k) res[miss[k]] = mr[k];
  return 0;
}

int ef_ahead_flush(ef_backend* be) {
  ef_ahead* ah = be->ahead;
  if (!ah || ah->n_pending == 0) return 0;
  const bool was = ah->collecting;
  ah->collecting = false;
  const int np = ah->n_pending;
  ef_dp_res r[EF_AHEAD_MAX];
  memset(r, 0, (size_t)np * sizeof(ef_dp_res));
  const int rc = ef_backend_ask(be, ah->q + ah->n, r, (size_t)np);
  if (rc == 0)
    for (int k = 0; k < np; ++k) ef_ahead_keep_next(ah, r[k].v);
  if (ef_prof_on) ef_prof.ahead_asked += (unsigned long long)np;
  ah->n_pending = 0;
  ah->collecting = was;
  return rc;
}

/* one question; 0 = answered, EF_DP_PENDING = noted (collect mode) */
static int run_dp(ef_backend* be, int kind, const char* a, size_t la, const char* b, size_t lb,
                  uint32_t p0, uint32_t p1, uint32_t p2, uint32_t tail, bool temp, ef_dp_res* res) {
  ef_dp_req rq = { kind, a, la, b, lb, p0, p1, p2, tail, temp ? 1u : 0u };
  const int rc = ef_dp_one(be, &rq, res);
  if (rc != 0 && rc != EF_DP_PENDING) {
    fprintf(stderr, "* FATAL dynamic-programming backend failed (kind %d, %zu x %zu)\n", kind, la, lb);
    abort();
  }
  return rc;
}

uint32_t ef_edit_distance(ef_backend* be, const char* a, size_t la, const char* b, size_t lb) {
  ef_dp_res r;
  run_dp(be, EF_DP_ED, a, la, b, lb, 0, 0, 0, 0, true, &r);      /* (the callers pass copies) */
  return (uint32_t)r.v[0];
}

uint32_t ef_compute_edit_distance(ef_backend* be, const char* a, size_t la, const char* b, size_t lb) {
  if (la == lb && strncmp(a, b, la) == 0) return 0;                 /* src/compute-alignments.c:242 */
  return ef_edit_distance(be, a, la, b, lb);
}

/* ---------------------------------------------------------------------------------------------- */
/* embeddings (src/est-factorizations.c:597-917, 1362-1460)                                       */
/* ---------------------------------------------------------------------------------------------- */
/* An embedding is a short sequence of (p, t, l) that is only ever copied, extended at its head and read
 * front to back: one block, e[0] = head (the reference keeps a linked list of heap records; here a
 * copy is a memcpy and a walk touches one or two cache lines instead of two per element). */
typedef struct { uint32_t n; ptl e[]; } emb;
static emb* emb_new(uint32_t n) { emb* x = (emb*)malloc(sizeof(emb) + (size_t)n * sizeof(ptl)); x->n = n; return x; }
static void embedding_free(void* e) { free(e); }

static emb* embedding_copy(const emb* e) {
  emb* c = emb_new(e->n);
  memcpy(c->e, e->e, (size_t)e->n * sizeof(ptl));
  return c;
}

/* update_embedding (:765-917): the embedding extended by `node` at its head, or NULL */
static emb* update_embedding(const emb* embedding, const ef_pairing* node, const char* GEN,
                             const ef_config* cfg) {
  const ptl* head = &embedding->e[0];
  if (head->p == EF_SINK_START) {
    if (node->p < 0) return NULL;
    emb* e = emb_new(1);
    e->e[0].p = node->p; e->e[0].t = node->t; e->e[0].l = node->l;
    return e;
  }
  if (node->p < 0) return embedding_copy(embedding);
  const int small_delta = (head->p + head->l) - node->p;
  const int big_delta = (head->t + head->l) - node->t;
  const int min_fl = (int)cfg->min_factor_len;
  const int fl = 2 * min_fl;
  if (!(small_delta >= fl && big_delta >= fl)) return NULL;
  if (!(small_delta - (node->l + head->l) <= fl)) return NULL;
  if (!(small_delta - big_delta <= fl)) return NULL;
  int head_l, head_p, head_t, node_l;
  if (small_delta >= (node->l + head->l) && big_delta >= (node->l + head->l)) {
    head_p = head->p; head_t = head->t; head_l = head->l; node_l = node->l;
  } else {
    const int ref_delta = small_delta < big_delta ? small_delta : big_delta;
    int len_node = ref_delta / 2;
    int len_head = ref_delta - len_node;
    if (len_node > node->l) { len_node = node->l; len_head = ref_delta - len_node; }
    else if (len_head > head->l) { len_head = head->l; len_node = ref_delta - len_head; }
    head_l = len_head;
    head_p = head->p + head->l - head_l;
    head_t = head->t + head->l - head_l;
    node_l = len_node;
  }
  const bool overlap_on_p = small_delta < (node->l + head->l);
  const int gap_on_p = head_p - node->p - node_l - 1;
  const int gap_on_t = head_t - node->t - node_l - 1;
  const int intron_len = gap_on_t - (gap_on_p > 0 ? gap_on_p : 0);
  const bool intron_on_t = intron_len >= 0 && (cfg->min_intron_length == 0 || intron_len >= cfg->min_intron_length);
  if (overlap_on_p && intron_on_t) {                       /* best cut by Burset frequency */
    int best_freq = -1, best_cut = 0;
    const int lo = (node->p + min_fl > head->p) ? node->p + min_fl : head->p;
    const int hi = (head->p + head->l - min_fl < node->p + node->l) ? head->p + head->l - min_fl : node->p + node->l;
    for (int cut = lo; cut <= hi; ++cut) {
      const int f = ef_burset_adaptor(GEN, (size_t)(cut - node->p + node->t), (size_t)(cut - head->p + head->t));
      if (f >= best_freq) { best_freq = f; best_cut = cut; }
    }
    const int dH = best_cut - head->p;
    head_l = head->l - dH; head_p = head->p + dH; head_t = head->t + dH;
    node_l = node->l - (node->p + node->l - best_cut);
  }
  if (!(gap_on_t <= fl || intron_on_t)) return NULL;
  emb* c = emb_new(embedding->n + 1);               /* the new head, the old head trimmed, the rest as it was */
  c->e[0].p = node->p; c->e[0].t = node->t; c->e[0].l = node_l;
  c->e[1].p = head_p; c->e[1].t = head_t; c->e[1].l = head_l;
  memcpy(c->e + 2, embedding->e + 1, (size_t)(embedding->n - 1) * sizeof(ptl));
  return c;
}

/* does x contain y (both on P and on T)?  used by maximality_relation */
static bool pair_inside(const ptl* inner, const ptl* outer) {
  if (inner->p < outer->p || (inner->p + inner->l > outer->p + outer->l)) return false;
  if (inner->t < outer->t || (inner->t + inner->l > outer->t + outer->l)) return false;
  return true;
}

/* maximality_relation (:1362-1460): 2 = add is maximal, 1 = both, 0 = cmp is maximal */
static int maximality_relation(const emb* add, const emb* cmp) {
  if (add->n > cmp->n) {
    bool check = true;
    for (uint32_t i = 0; i < cmp->n && check; ++i) check = pair_inside(&cmp->e[i], &add->e[i]);
    return check ? 2 : 1;
  }
  if (add->n < cmp->n) {
    bool check = true;
    for (uint32_t i = 0; i < add->n && check; ++i) check = pair_inside(&add->e[i], &cmp->e[i]);
    return check ? 0 : 1;
  }
  bool check = true;
  for (uint32_t i = 0; i < add->n && check; ++i) check = pair_inside(&add->e[i], &cmp->e[i]);
  if (check) return 0;
  check = true;
  for (uint32_t i = 0; i < add->n && check; ++i) check = pair_inside(&cmp->e[i], &add->e[i]);
  return check ? 2 : 1;
}

/* The reference bounds the enumeration with a wall-clock timeout (max_single_factorization_time,
 * checked at :626-629 on entering a subtree that is not memoised yet and at :706-711 every 1024
 * candidate embeddings); when it expires get_subtree_embeddings returns NULL and compute_est_fact
 * retries with a longer minimum factor (src/compute-est-fact.c:250-286).  Nothing here may depend
 * on time, so the same two places count WORK instead: one unit per subtree entered and per
 * candidate embedding compared.  The budget is max_single_factorization_time x EF_WORK_PER_SECOND
 * (what one host core gets through per second of this loop, rounded down to a power of ten);
 * PINTRON_WORK_BUDGET=<units> overrides it (tests).  The largest use of a run is reported by
 * ef_work_high_water(). */
#define EF_WORK_PER_SECOND 1000000ull
typedef struct { unsigned long long used, limit; } ef_work;
static unsigned long long work_high_water;
unsigned long long ef_work_high_water(void) { return __atomic_load_n(&work_high_water, __ATOMIC_RELAXED); }
extern unsigned long long ef_work_budget_override;      /* PINTRON_WORK_BUDGET, read once by ef_config_load (ef_config.c) */
static unsigned long long work_limit(const ef_config* cfg) {
  if (ef_work_budget_override) return ef_work_budget_override;
  return cfg->max_single_factorization_time ? (unsigned long long)cfg->max_single_factorization_time * EF_WORK_PER_SECOND : ~0ull;
}

/* get_subtree_embeddings (:597-762), memoised on the vertex; NULL = the work budget is spent */
static ef_list* subtree_embeddings(ef_pairing* root, const ef_config* cfg, const char* GEN, ef_work* wk) {
  if (root->emb_memo) return root->emb_memo;
  if (++wk->used > wk->limit) return NULL;
  ef_list* out = efl_new();
  root->visited = true;
  if (efl_empty(root->adjs)) {
    emb* e = emb_new(1);
    e->e[0].p = root->p; e->e[0].t = root->t; e->e[0].l = root->l;
    efl_push_front(out, e);
  } else {
    ef_iter ai = efl_begin(root->adjs);
    while (efi_has_next(&ai)) {
      ef_pairing* adj = (ef_pairing*)efi_next(&ai);
      ef_list* sub = subtree_embeddings(adj, cfg, GEN, wk);
      if (!sub) { efl_free(out, embedding_free); return NULL; }
      ef_iter si = efl_begin(sub);
      while (efi_has_next(&si)) {
        emb* add = update_embedding((const emb*)efi_next(&si), root, GEN, cfg);
        if (!add) continue;
        if (++wk->used > wk->limit) { embedding_free(add); efl_free(out, embedding_free); return NULL; }
        int is_max = 2;
        ef_iter ci = efl_begin(out);
        while (efi_has_next(&ci) && is_max >= 1) {
          const emb* cmp = (const emb*)efi_next(&ci);
          is_max = maximality_relation(add, cmp);
          if (is_max == 2) efi_remove(&ci, embedding_free);
        }
        if (is_max >= 1) efl_push_back(out, add); else embedding_free(add);
      }
    }
  }
  root->emb_memo = out;
  return out;
}

/* get_factorizations_from_embeddings (:1292-1356) */
static ef_list* factorizations_from_embeddings(ef_list* embeddings, const ef_config* cfg) {
  ef_list* out = efl_new();
  const int fl = 2 * (int)cfg->min_factor_len;
  ef_iter ei = efl_begin(embeddings);
  while (efi_has_next(&ei)) {
    const emb* em = (const emb*)efi_next(&ei);
    ef_list* fact = efl_new();
    ef_factor* last = NULL;
    for (uint32_t q = 0; q < em->n; ++q) {
      const ptl* x = &em->e[q];
      if (efl_empty(fact) || (x->t - last->GEN_end - 1) > fl) {
        last = factor_new(x->p, x->p + x->l - 1, x->t, x->t + x->l - 1);
        efl_push_back(fact, last);
      } else {
        last = (ef_factor*)efl_tail(fact);
        last->EST_end = x->p + x->l - 1;
        last->GEN_end = x->t + x->l - 1;
      }
    }
    efl_push_back(out, fact);
  }
  return out;
}

/* ---------------------------------------------------------------------------------------------- */
/* cleaning steps on one candidate factorization                                                  */
/* ---------------------------------------------------------------------------------------------- */
static bool not_source_sink(ef_list* fact, int est_len) {             /* :2111-2125 */
  if (efl_size(fact) > 1) return true;
  const ef_factor* h = (const ef_factor*)efl_head(fact);
  return !(h->EST_start < 0 || h->EST_start >= est_len);
}

static bool exon_start_end_ok(ef_list* fact) {                        /* :1989-2018 */
  int prev_e = -1, prev_g = -1;
  ef_iter it = efl_begin(fact);
  while (efi_has_next(&it)) {
    const ef_factor* x = (const ef_factor*)efi_next(&it);
    if (x->EST_start > x->EST_end || x->GEN_start > x->GEN_end) return false;
    if (x->EST_start < prev_e || x->GEN_start < prev_g) return false;
    prev_e = x->EST_end; prev_g = x->GEN_end;
  }
  return true;
}

/* handle_endpoints (:2127-2301): the first and the last exon are re-aligned and trimmed.  With two
 * or more exons the two alignments do not depend on each other and are requested together; with a
 * single exon the second alignment sees the trimmed exon, as in the reference. */
static void endpoint_head_apply(ef_list* fact, ef_factor* head, const ef_dp_res* r) {
  const char* ea = r->s0; const char* ga = r->s1;
  const int dim = r->v[1];
  int j = 0, matches = 0, cut_factor = head->EST_start, cut_exon = head->GEN_start;
  bool stop = false;
  while (j < dim && !stop) {
    if (matches > 5) stop = true;
    else {
      if (ea[j] == ga[j]) { ++cut_factor; ++cut_exon; ++matches; }
      else { if (ea[j] != '-') ++cut_factor; if (ga[j] != '-') ++cut_exon; matches = 0; }
      ++j;
    }
  }
  if (!stop) free(efl_pop_front(fact));
  else { head->EST_start = cut_factor - matches; head->GEN_start = cut_exon - matches; }
}

static void endpoint_tail_apply(ef_list* fact, ef_factor* tail, const ef_dp_res* r) {
  char* ea = r->s0; char* ga = r->s1;       /* zero-padded and ours to rewrite (ef_dp_res) */
  const int dim = r->v[1];
  int j = dim - 1, matches = 0, cut_factor = tail->EST_end, cut_exon = tail->GEN_end;
  bool stop = false;
  while (j >= 0 && !stop) {
    if (matches > 10) stop = true;
    else {
      if (ea[j] == ga[j]) { --cut_factor; --cut_exon; ++matches; }
      else { if (ea[j] != '-') --cut_factor; if (ga[j] != '-') --cut_exon; matches = 0; }
      --j;
    }
  }
  int est_cleavage = cut_factor + matches, gen_cleavage = cut_exon + matches;
  int cursor = j + matches + 1;
  stop = false;
  while (((ea[cursor] == '-' || ga[cursor] == '-') && cursor < dim - 1) && !stop) {
    if (ea[cursor] == '-') {
      int tr = cursor + 1;
      while (ea[tr] == '-') ++tr;
      if (tr < dim && ea[tr] == ga[cursor]) { ea[cursor] = ea[tr]; ea[tr] = '-'; ++est_cleavage; ++gen_cleavage; }
      else stop = true;
    } else {
      int tr = cursor + 1;
      while (ga[tr] == '-') ++tr;
      if (tr < dim && ga[tr] == ea[cursor]) { ga[cursor] = ga[tr]; ga[tr] = '-'; ++est_cleavage; ++gen_cleavage; }
      else stop = true;
    }
    ++cursor;
  }
  if (gen_cleavage >= tail->GEN_start) { tail->EST_end = est_cleavage; tail->GEN_end = gen_cleavage; }
  else free(efl_pop_back(fact));
}

static void endpoint_request(ef_dp_req* q, const ef_factor* x, const char* gen, const char* est) {
  const ef_dp_req r = { EF_DP_ALIGN, est + x->EST_start, view_len(est, x->EST_start, x->EST_end - x->EST_start + 1),
                        gen + x->GEN_start, view_len(gen, x->GEN_start, x->GEN_end - x->GEN_start + 1), 0, 0, 0, 0, 0 };
  *q = r;
}
static void endpoint_release(ef_dp_req* q, ef_dp_res* r) {
  (void)q;
  ef_dp_res_release(r);
}

static unsigned max_edit_for_exon(size_t exon_length);
/* The "exon check" of one exon (include/pintron_gpu.h: KBAND with tail = 1): the banded edit distance clean_noisy_exons
 * asks for (:1842-1898) and, beside it, the two dust-score comparisons of clean_low_complexity_exons_2 (:1667-1704) --
 * both look at the same two strings, the exon on the genomic sequence and on the EST.  false: the exon is empty on
 * the genomic sequence (neither routine looks at it). */
static bool exon_check_request(ef_dp_req* q, const ef_factor* x, const char* gen, const char* est, const ef_config* cfg) {
  if (x->GEN_start > x->GEN_end) return false;
  uint32_t w[2];
  memcpy(w, &cfg->complexity_threshold, 8);
  const ef_dp_req r = { EF_DP_KBAND, gen + x->GEN_start, view_len(gen, x->GEN_start, x->GEN_end - x->GEN_start + 1),
                        est + x->EST_start, view_len(est, x->EST_start, x->EST_end - x->EST_start + 1),
                        max_edit_for_exon((size_t)(x->GEN_end - x->GEN_start + 1)), w[0], w[1], 1, 0 };
  *q = r;
  return true;
}

static ef_list* handle_endpoints(ef_list* fact, const char* gen, const char* est, const ef_config* cfg, ef_backend* be) {
  ef_factor* head = (ef_factor*)efl_head(fact);
  ef_dp_req q[2]; ef_dp_res r[2];
  if (efl_size(fact) >= 2) {
    ef_factor* tail = (ef_factor*)efl_tail(fact);
    /* the exon checks of the INTERNAL exons (their coordinates are final here) go out with the two alignments; their
     * answers are kept for the cleaning steps that follow (estfact.h: ef_ahead) */
    enum { INNER_MAX = 22 };
    ef_dp_req qq[2 + INNER_MAX]; ef_dp_res rr[2 + INNER_MAX]; size_t nq = 2;
    endpoint_request(&qq[0], head, gen, est);
    endpoint_request(&qq[1], tail, gen, est);
    if (be->ahead && ef_endpoint_checks) {
      /* p0 = 1 / 2: "this is a first / last exon" -- a backend may answer, beside the alignment, the exon check of
       * the exon as handle_endpoints is going to trim it (include/pintron_gpu.h: ALIGN with p0); p1, p2 = the
       * complexity threshold's double bits, as in the exon check itself */
      uint32_t w[2];
      memcpy(w, &cfg->complexity_threshold, 8);
      qq[0].p0 = 1; qq[0].p1 = w[0]; qq[0].p2 = w[1];
      qq[1].p0 = 2; qq[1].p1 = w[0]; qq[1].p2 = w[1];
    }
    if (be->ahead && efl_size(fact) >= 3 && efl_size(fact) <= INNER_MAX) {
      ef_iter it = efl_begin(fact);
      efi_next(&it);
      while (efi_has_next(&it)) {
        const ef_factor* x = (const ef_factor*)efi_next(&it);
        if (x == tail) break;
        if (exon_check_request(&qq[nq], x, gen, est, cfg)) ++nq;
      }
    }
    if (ef_dp_many(be, qq, rr, nq) != 0) { fprintf(stderr, "* FATAL alignment backend failed\n"); abort(); }
    for (size_t k = 2; k < nq; ++k) ef_ahead_put(be->ahead, &qq[k], rr[k].v);
    if (be->ahead) {
      /* an exon check that came with an alignment is filed under the question it answers: the exon as the backend
       * trimmed it (v[2], v[3]) with the bound it used (v[4] >> 8).  The trimming below is the host's own, on the
       * strings; the cleaning step finds the answer only if it then asks exactly this question. */
      for (int k = 0; k < 2; ++k) {
        const int32_t* v = rr[k].v;
        if (!(v[4] & 8)) continue;
        const ef_factor* x = k == 0 ? head : tail;
        ef_factor y = *x;
        if (k == 0) { y.EST_start += v[2]; y.GEN_start += v[3]; }
        else { y.EST_end = y.EST_start + v[2] - 1; y.GEN_end = y.GEN_start + v[3] - 1; }
        ef_dp_req cq;
        if (y.EST_start > y.EST_end || !exon_check_request(&cq, &y, gen, est, cfg) || cq.p0 != ((uint32_t)v[4] >> 8)) continue;
        /* (the operands of that question must be the very stretches the backend looked at) */
        if (cq.la != (size_t)(y.GEN_end - y.GEN_start + 1) || cq.lb != (size_t)(y.EST_end - y.EST_start + 1)) continue;
        const int32_t answer[6] = { v[4] & 1, 0, (v[4] >> 1) & 3, 0, 0, 0 };
        ef_ahead_put(be->ahead, &cq, answer);
        if (ef_prof_on) ++ef_prof.ahead_asked;
      }
    }
    endpoint_head_apply(fact, head, &rr[0]);
    endpoint_tail_apply(fact, tail, &rr[1]);     /* `tail` is not the exon the head step touched */
    endpoint_release(&qq[0], &rr[0]); endpoint_release(&qq[1], &rr[1]);
    return fact;
  }
  endpoint_request(&q[0], head, gen, est);
  if (ef_dp_many(be, q, r, 1) != 0) { fprintf(stderr, "* FATAL alignment backend failed\n"); abort(); }
  endpoint_head_apply(fact, head, &r[0]);
  endpoint_release(&q[0], &r[0]);
  if (efl_empty(fact)) return fact;
  ef_factor* tail = (ef_factor*)efl_tail(fact);
  endpoint_request(&q[0], tail, gen, est);
  if (ef_dp_many(be, q, r, 1) != 0) { fprintf(stderr, "* FATAL alignment backend failed\n"); abort(); }
  endpoint_tail_apply(fact, tail, &r[0]);
  endpoint_release(&q[0], &r[0]);
  return fact;
}

static bool is_ch(char c, char up) { return c == up || c == (char)(up + 32); }

/* clean_external_exons (:1706-1826) */
ef_list* ef_clean_external_exons(ef_list* fact, const char* gen, const char* est, ef_backend* be) {
  if (efl_empty(fact)) return fact;
  ef_factor* head = (ef_factor*)efl_pop_front(fact);
  const int hl = head->GEN_end - head->GEN_start + 1;
  bool ok = hl >= 10;
  if (ok && hl < 20) {
    if (!is_ch(gen[head->GEN_end + 1], 'G')) ok = false;
    else if (!is_ch(gen[head->GEN_end + 2], 'T') && !is_ch(gen[head->GEN_end + 2], 'C')) ok = false;
    else if (efl_size(fact) >= 1) {
      const ef_factor* nx = (const ef_factor*)efl_head(fact);
      if (!is_ch(gen[nx->GEN_start - 2], 'A')) ok = false;
      else if (!is_ch(gen[nx->GEN_start - 1], 'G')) ok = false;
    } else ok = false;
    if (ok) {
      char* g = ef_real_substring(head->GEN_start, head->GEN_end - head->GEN_start + 1, gen);
      char* e = ef_real_substring(head->EST_start, head->EST_end - head->EST_start + 1, est);
      if ((int)ef_edit_distance(be, g, strlen(g), e, strlen(e)) > 0) ok = false;
      free(g); free(e);
    }
  }
  if (ok) efl_push_front(fact, head); else free(head);
  if (efl_empty(fact)) return fact;
  ef_factor* tail = (ef_factor*)efl_pop_back(fact);
  const int tl = tail->GEN_end - tail->GEN_start + 1;
  ok = tl >= 10;
  if (ok && tl < 20) {
    if (!is_ch(gen[tail->GEN_start - 2], 'A')) ok = false;
    else if (!is_ch(gen[tail->GEN_start - 1], 'G')) ok = false;
    else if (efl_size(fact) >= 1) {
      const ef_factor* pv = (const ef_factor*)efl_tail(fact);
      if (!is_ch(gen[pv->GEN_end + 1], 'G')) ok = false;
      else if (!is_ch(gen[pv->GEN_end + 2], 'T') && !is_ch(gen[pv->GEN_end + 2], 'C')) ok = false;
    } else ok = false;
    if (ok) {
      char* g = ef_real_substring(tail->GEN_start, tail->GEN_end - tail->GEN_start + 1, gen);
      char* e = ef_real_substring(tail->EST_start, tail->EST_end - tail->EST_start + 1, est);
      if ((int)ef_edit_distance(be, g, strlen(g), e, strlen(e)) > 0) ok = false;
      free(g); free(e);
    }
  }
  if (ok) efl_push_back(fact, tail); else free(tail);
  return fact;
}

/* update_with_subfact_with_best_coverage (:1900-1987): split_idx = 1-based indices of bad exons */
static ef_list* keep_best_run(ef_list* fact, const int* split_idx, int n_split) {
  if (n_split == 0) return fact;
  int best_l = -1, best_r = -1, best_cover = -1;
  ef_iter fi = efl_begin(fact);
  int left = 1;
  for (int k = 0; k < n_split; ++k) {
    const int right = split_idx[k];
    const ef_factor* le = (const ef_factor*)efi_next(&fi);
    const ef_factor* re = le;
    if (left < right) {
      for (int times = right - left - 1; times > 0; --times) re = (const ef_factor*)efi_next(&fi);
      const int cover = re->EST_end - le->EST_start + 1;
      if (cover > best_cover) { best_l = left; best_r = right - 1; best_cover = cover; }
      efi_next(&fi);
    }
    left = right + 1;
  }
  const int size = (int)efl_size(fact);
  if (left <= size) {
    const ef_factor* le = (const ef_factor*)efi_next(&fi);
    const ef_factor* re = le;
    for (int times = size - left; times > 0; --times) re = (const ef_factor*)efi_next(&fi);
    const int cover = re->EST_end - le->EST_start + 1;
    if (cover > best_cover) { best_l = left; best_r = size; best_cover = cover; }
  }
  if (best_l == -1 || best_r == -1) {
    for (int k = size; k > 0; --k) free(efl_pop_back(fact));
  } else {
    for (int k = best_l - 1; k > 0; --k) free(efl_pop_front(fact));
    for (int k = best_r + 1; k <= size; ++k) free(efl_pop_back(fact));
  }
  return fact;
}

static unsigned max_edit_for_exon(size_t exon_length) {                 /* :1828-1840 */
  const double rate = exon_length > 100 ? 0.030 : (exon_length > 50 ? 0.035 : 0.040);
  const double c = ceil(exon_length * rate);
  return (unsigned)(c > 1.0 ? c : 1.0);
}

/* clean_noisy_exons (:1842-1898): the banded edit distances of the exons do not depend on each
 * other, so they are requested together */
ef_list* ef_clean_noisy_exons(ef_list* fact, const char* gen, const char* est, bool only_internals, ef_backend* be) {
  const size_t size = efl_size(fact);
  /* factorizations have a handful of exons: the work arrays live on the (fibre) stack */
  enum { SMALL = 24 };
  int idx_small[SMALL], slot_small[SMALL]; ef_dp_req rq_small[SMALL]; ef_dp_res rs_small[SMALL];
  const bool small = size < SMALL;
  int* idx = small ? idx_small : (int*)malloc((size + 1) * sizeof(int));
  ef_dp_req* rq = small ? rq_small : (ef_dp_req*)malloc((size + 1) * sizeof(ef_dp_req));
  ef_dp_res* rs = small ? rs_small : (ef_dp_res*)malloc((size + 1) * sizeof(ef_dp_res));
  int* slot = small ? slot_small : (int*)malloc((size + 1) * sizeof(int));   /* request of each visited exon, -1 = none */
  size_t nrq = 0, nvis = 0;
  int index = only_internals ? 2 : 1;
  const int first_index = index;
  const int last = only_internals ? (int)(size - 1) : (int)size;
  ef_iter it = efl_begin(fact);
  if (only_internals) efi_next(&it);
  while (efi_has_next(&it) && index <= last) {
    const ef_factor* x = (const ef_factor*)efi_next(&it);
    const unsigned max_err = max_edit_for_exon((size_t)(x->GEN_end - x->GEN_start + 1));
    slot[nvis] = -1;
    if (x->GEN_start <= x->GEN_end) {
      /* the exon on the genomic sequence and on the EST, read in place (the reference copies them) */
      const ef_dp_req q = { EF_DP_KBAND, gen + x->GEN_start, view_len(gen, x->GEN_start, x->GEN_end - x->GEN_start + 1),
                            est + x->EST_start, view_len(est, x->EST_start, x->EST_end - x->EST_start + 1), max_err, 0, 0, 0, 0 };
      slot[nvis] = (int)nrq;
      rq[nrq++] = q;
    }
    ++nvis; ++index;
  }
  if (ef_dp_many(be, rq, rs, nrq) != 0) { fprintf(stderr, "* FATAL dynamic-programming backend failed (K-band)\n"); abort(); }
  int n = 0;
  for (size_t v = 0; v < nvis; ++v) {
    const bool ok = slot[v] >= 0 && rs[slot[v]].v[0] != 0;
    if (!ok) idx[n++] = first_index + (int)v;
  }
  fact = keep_best_run(fact, idx, n);
  if (!small) { free(idx); free(rq); free(rs); free(slot); }
  return fact;
}

static bool est_coverage_ok(ef_list* fact, const char* est) {           /* :2303-2321 */
  const size_t len = strlen(est);
  const ef_factor* h = (const ef_factor*)efl_head(fact);
  const ef_factor* t = (const ef_factor*)efl_tail(fact);
  const double cov = (double)(t->EST_end - h->EST_start + 1) / (double)len;
  return cov >= 0.35f;
}

/* ---------------------------------------------------------------------------------------------- */
/* relaxed containment of factorizations (src/est-factorizations.c:1159-1259, src/list.c:306-483)  */
/* ---------------------------------------------------------------------------------------------- */
static int relaxed_factor_compare(const ef_factor* p1, const ef_factor* p2, int cfr_type, int allowed, ef_list* l1) {
  if (p1->GEN_start < p2->GEN_start && p1->GEN_end < p2->GEN_start) return 1;
  if (p2->GEN_start < p1->GEN_start && p2->GEN_end < p1->GEN_start) return 1;
  const int max_unconf = 20;
  if (cfr_type == 0) {
    if (abs(p1->GEN_end - p2->GEN_end) <= allowed && abs(p1->GEN_start - p2->GEN_start) <= allowed) return 0;
  }
  if (abs(cfr_type) == 2) {
    if (abs(p1->GEN_end - p2->GEN_end) <= allowed) {
      if (cfr_type == 2) {
        if (p1->GEN_start - p2->GEN_start > max_unconf) return 1;
        if (p1->GEN_start - p2->GEN_start > 0) {
          int tot = 0; bool stop = false;
          ef_iter it = efl_begin(l1);
          while (efi_has_next(&it) && !stop) {
            const ef_factor* f = (const ef_factor*)efi_next(&it);
            if (p1->GEN_start == f->GEN_start) stop = true; else tot += f->GEN_end - f->GEN_start + 1;
          }
          if (abs(p1->GEN_start - p2->GEN_start - tot) < 10) return 1;
        }
      }
      return 0;
    }
  }
  if (abs(cfr_type) == 1) {
    if (abs(p1->GEN_start - p2->GEN_start) <= allowed) {
      if (cfr_type == 1) {
        if (p2->GEN_end - p1->GEN_end > max_unconf) return 1;
        if (p2->GEN_end - p1->GEN_end > 0) {
          int tot = 0; bool stop = false;
          ef_iter it = efl_end(l1);
          while (efi_has_prev(&it) && !stop) {
            const ef_factor* f = (const ef_factor*)efi_prev(&it);
            if (p1->GEN_start == f->GEN_start) stop = true; else tot += f->GEN_end - f->GEN_start + 1;
          }
          if (abs(p2->GEN_end - p1->GEN_end - tot) < 20) return 1;
        }
      }
      return 0;
    }
  }
  return 1;
}

/* relaxed_list_compare (src/list.c:449-483): -2 when equal, else 0 */
static int relaxed_list_compare(ef_list* l1, ef_list* l2, int allowed_diff) {
  if (efl_size(l1) != efl_size(l2) || efl_size(l1) == 1) return 0;
  ef_iter i1 = efl_begin(l1), i2 = efl_begin(l2);
  const int actual = allowed_diff == -1 ? 0 : allowed_diff;
  int count = 1;
  while (efi_has_next(&i1) && efi_has_next(&i2)) {
    const int type = allowed_diff == -1 ? 0 : (count == 1 ? -2 : (count == (int)efl_size(l1) ? -1 : 0));
    if (relaxed_factor_compare((const ef_factor*)i1.next->el, (const ef_factor*)i2.next->el, type, actual, l1) != 0) return 0;
    efi_next(&i1); efi_next(&i2);
    ++count;
  }
  return -2;
}

/* relaxed_list_contained (src/list.c:320-441): 0 different, -1 l1 in l2, 1 l2 in l1, -2 equal */
static int relaxed_list_contained(ef_list* l1, ef_list* l2, int allowed_diff) {
  if (efl_size(l1) == efl_size(l2)) return relaxed_list_compare(l1, l2, allowed_diff);
  if (efl_size(l1) == 1 || efl_size(l2) == 1) return 0;
  const int actual = allowed_diff == -1 ? 0 : allowed_diff;
  ef_iter i1 = efl_begin(l1), i2 = efl_begin(l2);
  bool found = false;
  int type = allowed_diff == -1 ? 0 : -2;
  unsigned count_long = 1;
  const bool l1_longer = efl_size(l1) > efl_size(l2);
  ef_list* lng = l1_longer ? l1 : l2;
  ef_iter* il = l1_longer ? &i1 : &i2;      /* iterator on the longer list */
  ef_iter* is = l1_longer ? &i2 : &i1;
  while (efi_has_next(il) && !found) {
    if (relaxed_factor_compare((const ef_factor*)il->next->el, (const ef_factor*)is->next->el, type, actual, lng) == 0) { found = true; efi_next(is); }
    else ++count_long;
    efi_next(il);
    if (type == -2) type = 2;
  }
  if (!found) return 0;
  unsigned count_factors = 1;
  bool stop = false;
  const size_t short_size = l1_longer ? efl_size(l2) : efl_size(l1);
  const size_t long_size = l1_longer ? efl_size(l1) : efl_size(l2);
  while (efi_has_next(&i1) && efi_has_next(&i2) && !stop) {
    type = allowed_diff == -1 ? 0 : ((count_factors + 1 == short_size) ? ((count_long + 1 == long_size) ? -1 : 1) : 0);
    if (relaxed_factor_compare((const ef_factor*)il->next->el, (const ef_factor*)is->next->el, type, actual, lng) == 0) { efi_next(&i1); efi_next(&i2); }
    else stop = true;
    ++count_factors; ++count_long;
  }
  if (stop) return 0;
  if (efl_size(l1) >= efl_size(l2)) return count_factors == efl_size(l2) ? 1 : 0;
  return count_factors == efl_size(l1) ? -1 : 0;
}

/* add_if_not_exists (:2041-2109) */
ef_list* ef_add_if_not_exists(ef_list* to_add, ef_list* list, const ef_config* cfg, bool* added) {
  ef_iter ci = efl_begin(list);
  bool found = false;
  while (efi_has_next(&ci) && !found) {
    ef_list* cmp = (ef_list*)efi_next(&ci);
    int res = 0;
    if (efl_size(cmp) == efl_size(to_add) && efl_size(cmp) == 1) {
      const ef_factor* h1 = (const ef_factor*)efl_head(to_add); const ef_factor* h2 = (const ef_factor*)efl_head(cmp);
      if (h1->GEN_start == h2->GEN_start && h1->GEN_end == h2->GEN_end) res = -2;
      else if (h1->GEN_start >= h2->GEN_start && h1->GEN_end <= h2->GEN_end) res = -1;
      else if (h1->GEN_start <= h2->GEN_start && h1->GEN_end >= h2->GEN_end) res = 1;
    } else {
      res = relaxed_list_contained(to_add, cmp, (int)cfg->max_site_difference);
    }
    if (res < 0) {
      if (res == -2) {
        const ef_factor* h1 = (const ef_factor*)efl_head(to_add); ef_factor* h2 = (ef_factor*)efl_head(cmp);
        if (h1->EST_start < h2->EST_start) { h2->EST_start = h1->EST_start; h2->GEN_start = h1->GEN_start; }
        const ef_factor* t1 = (const ef_factor*)efl_tail(to_add); ef_factor* t2 = (ef_factor*)efl_tail(cmp);
        if (t1->EST_end > t2->EST_end) { t2->EST_end = t1->EST_end; t2->GEN_end = t1->GEN_end; }
      }
      found = true;
    } else if (res == 1) {
      efi_remove(&ci, ef_factorization_free);
    }
  }
  if (!found) efl_push_back(list, to_add);
  *added = !found;
  return list;
}

/* ---------------------------------------------------------------------------------------------- */
/* filters                                                                                        */
/* ---------------------------------------------------------------------------------------------- */
static double coverage_of(ef_list* fact, unsigned length) {             /* :1262-1273 */
  const ef_factor* h = (const ef_factor*)efl_head(fact); const ef_factor* t = (const ef_factor*)efl_tail(fact);
  const int cover = (int)length - (h->EST_start + ((int)length - t->EST_end - 1));
  return ((double)cover) / (double)length;
}

static int gap_length_of(ef_list* fact) {                               /* :1276-1290 */
  if (efl_size(fact) == 1) return 0;
  int g = 0;
  ef_iter it = efl_begin(fact);
  const ef_factor* d = (const ef_factor*)efi_next(&it);
  while (efi_has_next(&it)) { const ef_factor* a = (const ef_factor*)efi_next(&it); g += a->EST_start - d->EST_end - 1; d = a; }
  return g;
}

/* check_gap_errors (:1462-1546) */
static bool check_gap_errors(ef_list* fact, const char* est, const char* gen, const ef_config* cfg, ef_backend* be) {
  (void)cfg;
  const unsigned threshold_ed = 20;
  unsigned tot = 0;
  bool ok = true;
  ef_iter it = efl_begin(fact);
  ef_factor* d = (ef_factor*)efi_next(&it);
  while (efi_has_next(&it) && ok) {
    ef_factor* a = (ef_factor*)efi_next(&it);
    const size_t gapP = (size_t)(a->EST_start - d->EST_end - 1);
    if (gapP > 0) {
      const size_t gapT = (size_t)(a->GEN_start - d->GEN_end - 1);
      const int max_errs = (int)gapP;
      /* refine_borders(p, gapP, t, gapT, max_errs) on fresh NUL-terminated copies in the reference
       * => tail 0: nothing beyond the two windows is looked at, so windows that lie inside the
       * sequences are passed where they are (the genomic gap is a whole intron: no copy, and
       * the GPU reads it from the resident genomic) */
      char *pc = NULL, *tc = NULL;
      const char* p = est + d->EST_end + 1;
      const char* t = gen + d->GEN_end + 1;
      if (d->EST_end + 1 < 0 || view_len(est, d->EST_end + 1, (int)gapP) != gapP) p = pc = ef_real_substring(d->EST_end + 1, (int)gapP, est);
      if (d->GEN_end + 1 < 0 || (int)gapT < 0 || (size_t)(d->GEN_end + 1) + gapT > ef_genomic_len(gen)) t = tc = ef_real_substring(d->GEN_end + 1, (int)gapT, gen);
      ef_dp_res r;
      const int asked = run_dp(be, EF_DP_BORDERS, p, gapP, t, gapT, 0, (uint32_t)gapP, (uint32_t)max_errs, 0, pc || tc, &r);
      free(pc); free(tc);
      if (asked == EF_DP_PENDING) { d = a; continue; }           /* collect mode: noted; the next gap does not depend on this one */
      if (ef_collecting(be)) { d = a; continue; }                /* ... known already: nothing is changed in this mode */
      ok = r.v[0] != 0;
      if (ok) {
        tot += (unsigned)r.v[4];
        d->EST_end += r.v[1];
        a->EST_start = d->EST_end + 1;
        d->GEN_end += r.v[2];
        a->GEN_start -= (int)gapT - r.v[3];
      }
    }
    d = a;
  }
  if (ef_collecting(be)) return true;
  if (ok && tot > threshold_ed) ok = false;
  if (ok) {
    it = efl_begin(fact);
    d = (ef_factor*)efi_next(&it);
    while (efi_has_next(&it)) {
      ef_factor* a = (ef_factor*)efi_next(&it);
      if (a->GEN_start - d->GEN_end - 1 <= 3) { d->EST_end = a->EST_end; d->GEN_end = a->GEN_end; efi_remove(&it, free); }
      else d = a;
    }
  }
  return ok;
}

/* correct_composition_tail + detect_polyA_signal (src/detect-polya.c:42-164) */
static void correct_tail(ef_list* fact, const char* gen, const char* est) {
  ef_factor* tail = (ef_factor*)efl_tail(fact);
  size_t i = (size_t)(tail->EST_end + 1), j = (size_t)(tail->GEN_end + 1);
  const size_t el = strlen(est), gl = ef_genomic_len(gen);
  while (i < el && j < gl && gen[j] == est[i]) { ++i; ++j; }
  tail->EST_end = (int)i - 1; tail->GEN_end = (int)j - 1;
}

static bool detect_polyA(ef_list* fact, const char* gen, const char* est, bool* polyadenil) {
  const ef_factor* tail = (const ef_factor*)efl_tail(fact);
  const size_t el = strlen(est);
  const char* cl = est + tail->EST_end + 1;
  const int cll = (tail->EST_end + 1 <= (int)el) ? (int)(el - (size_t)tail->EST_end - 1) : 0;
  int i = 0, matches = 0;
  bool stop = false;
  while (i < cll && !stop) {
    if (cl[i] == 'a' || cl[i] == 'A') { if (matches >= 8) stop = true; else { ++matches; ++i; } }
    else { if (matches >= 8) stop = true; else i = cll; }
  }
  *polyadenil = false;
  if (stop) {
    i = tail->GEN_end - 39 > 0 ? tail->GEN_end - 39 : 0;
    while (i <= tail->GEN_end && !*polyadenil) {
      if (gen[i] == 'a' || gen[i] == 'A') {
        char pas[8];                                 /* real_substring(i, 6, gen): stops at the terminator */
        const size_t pl = view_len(gen, i, 6);
        memcpy(pas, gen + i, pl); pas[pl] = '\0';
        if (!strcmp(pas, "aataaa") || !strcmp(pas, "AATAAA") || !strcmp(pas, "attaaa") || !strcmp(pas, "ATTAAA")) *polyadenil = true;
      }
      ++i;
    }
    i = tail->GEN_end - 9 > 0 ? tail->GEN_end - 9 : 0;
    matches = 0;
    while (i <= tail->GEN_end + 10 && stop && gen[i] != '\0') {
      if (matches >= 6) stop = false;
      else { if (gen[i] == 'a' || gen[i] == 'A') ++matches; else matches = 0; ++i; }
    }
    if (stop) {
      i = tail->GEN_end + 1;
      int count = 0;
      while (i <= tail->GEN_end + 10 && stop && gen[i] != '\0') {
        if (count >= 7) stop = false;
        else { if (gen[i] == 'a' || gen[i] == 'A') ++count; ++i; }
      }
    }
  }
  return stop;
}

/* ---------------------------------------------------------------------------------------------- */
/* get_EST_factorizations (src/est-factorizations.c:126-594)                                       */
/* ---------------------------------------------------------------------------------------------- */
/* clean_low_complexity_exons_2 (:1667-1704) and clean_noisy_exons (:1842-1898) of one candidate, from ONE request: the
 * exon check of every exon (dust comparisons + banded distance, computed side by side on the device).  The first
 * routine keeps the best run of exons that are not low-complexity, the second -- on what is left -- the best run of
 * exons within their error bound; dropping exons does not change the others' strings, so every exon's answers stay
 * valid. */
static ef_list* clean_complexity_and_noise(ef_list* fact, const char* gen, const char* est, const ef_config* cfg, ef_backend* be) {
  const size_t size = efl_size(fact);
  enum { SMALL = 24 };
  const ef_factor* x_small[SMALL]; int slot_small[SMALL], idx_small[SMALL]; ef_dp_req rq_small[SMALL]; ef_dp_res rs_small[SMALL];
  const bool small = size < SMALL;
  const ef_factor** xs = small ? x_small : (const ef_factor**)malloc((size + 1) * sizeof(ef_factor*));
  int* slot = small ? slot_small : (int*)malloc((size + 1) * sizeof(int));
  int* idx = small ? idx_small : (int*)malloc((size + 1) * sizeof(int));
  ef_dp_req* rq = small ? rq_small : (ef_dp_req*)malloc((size + 1) * sizeof(ef_dp_req));
  ef_dp_res* rs = small ? rs_small : (ef_dp_res*)malloc((size + 1) * sizeof(ef_dp_res));
  size_t nx = 0, nrq = 0;
  ef_iter it = efl_begin(fact);
  while (efi_has_next(&it)) {
    const ef_factor* x = (const ef_factor*)efi_next(&it);
    xs[nx] = x;
    slot[nx] = exon_check_request(&rq[nrq], x, gen, est, cfg) ? (int)nrq++ : -1;
    ++nx;
  }
  if (ef_dp_many(be, rq, rs, nrq) != 0) { fprintf(stderr, "* FATAL dynamic-programming backend failed (exon checks)\n"); abort(); }
  /* low complexity: an exon that is empty on the genomic sequence scores 0 on both sides (:1683-1690) */
  int n = 0;
  for (size_t v = 0; v < nx; ++v) if (slot[v] >= 0 && rs[slot[v]].v[2] != 0) idx[n++] = (int)v + 1;
  fact = keep_best_run(fact, idx, n);
  if (!efl_empty(fact)) {
    ef_phase(EFP_NOISY);
    /* noisy exons among the survivors (a contiguous run of the exons above): found again by address */
    n = 0;
    int index = 1;
    size_t v = 0;
    it = efl_begin(fact);
    while (efi_has_next(&it)) {
      const ef_factor* x = (const ef_factor*)efi_next(&it);
      while (v < nx && xs[v] != x) ++v;
      const bool ok = v < nx && slot[v] >= 0 && rs[slot[v]].v[0] != 0;
      if (!ok) idx[n++] = index;
      ++index;
    }
    fact = keep_best_run(fact, idx, n);
  }
  if (!small) { free(xs); free(slot); free(idx); free(rq); free(rs); }
  return fact;
}

/* the cleaning steps of the candidate factorizations of one root (:203-262); the survivors join flist */
static ef_list* clean_candidates(ef_list* cands, ef_list* flist, unsigned est_len, const char* GEN, const char* EST,
                                 const ef_config* cfg, ef_backend* be) {
  ef_iter ci = efl_begin(cands);
  while (efi_has_next(&ci)) {
    ef_list* f = (ef_list*)efi_next(&ci);
    bool ok = not_source_sink(f, (int)est_len);
    if (ok) ok = exon_start_end_ok(f);
    if (ok) { ef_phase(EFP_ENDPOINTS); f = handle_endpoints(f, GEN, EST, cfg, be); if (efl_empty(f)) ok = false; }
    if (ok) { ef_phase(EFP_EXTERNAL); f = ef_clean_external_exons(f, GEN, EST, be); if (efl_empty(f)) ok = false; }
    if (ok) { ef_phase(EFP_DUST); f = clean_complexity_and_noise(f, GEN, EST, cfg, be); if (efl_empty(f)) ok = false; }
    ef_phase(EFP_ADD);
    if (ok) ok = est_coverage_ok(f, EST);
    if (ok) {
      bool added;
      flist = ef_add_if_not_exists(f, flist, cfg, &added);
      if (!added) ef_factorization_free(f);
    } else {
      ef_factorization_free(f);
    }
  }
  efl_free(cands, NULL);
  return flist;
}

/* A MEG that is ONE PATH from the source through every vertex to the sink -- what an EST that maps to one place
 * gives, nine times out of ten -- has exactly one embedding per vertex, and the enumeration of
 * get_subtree_embeddings (:597-762) from its first root, the source, comes down to folding update_embedding over
 * the path from the sink backwards; every vertex is visited by that, so there is no other root.  The path is read
 * off the device's record (vertices in position-list order + CSR, include/pintron_gpu.h) without building the
 * list structure.  Returns false when the graph is not such a path (the caller enumerates as usual); *out = the
 * source's embedding, or NULL when some vertex has none (update_embedding refused: no candidate at all). */
static bool chain_embedding(const void* rec, const ef_config* cfg, const char* GEN, ef_work* wk, emb** out) {
  const uint32_t* head = (const uint32_t*)rec;
  const uint32_t nv = head[0];
  if (nv < 2 || nv > 64) return false;
  const int32_t* vt = (const int32_t*)((const char*)rec + 16);
  const uint16_t* first = (const uint16_t*)((const char*)rec + 16 + 12 * (size_t)nv);
  const uint8_t* tgt = (const uint8_t*)rec + 16 + 12 * (size_t)nv + 2 * ((size_t)nv + 1);
  if (vt[0] != EF_SOURCE_START) return false;
  uint8_t path[64];
  uint64_t seen = 0;
  uint32_t k = 0, n = 0;
  for (;;) {                                           /* follow the only out-edge from the source */
    if (seen >> k & 1u) return false;
    seen |= 1ull << k; path[n++] = (uint8_t)k;
    const uint32_t deg = (uint32_t)first[k + 1] - first[k];
    if (deg == 0) break;
    if (deg != 1) return false;
    k = tgt[first[k]];
    if (k >= nv) return false;
  }
  if (n != nv || vt[3 * (size_t)path[n - 1]] != EF_SINK_START) return false;      /* every vertex, ending in the sink */
  /* the work the enumeration would have counted: one unit per vertex entered, one per embedding that comes back from
   * update_embedding (:626-629, :706-711); over the budget = "timeout expired", which the caller turns into the
   * reference's retry with longer factors */
  wk->used += nv;
  ef_pairing node;
  memset(&node, 0, sizeof node);
  emb* e = emb_new(1);
  { const int32_t* s3 = vt + 3 * (size_t)path[n - 1]; e->e[0].p = s3[0]; e->e[0].t = s3[1]; e->e[0].l = s3[2]; }
  for (uint32_t q = n - 1; q-- > 0 && e;) {
    const int32_t* v3 = vt + 3 * (size_t)path[q];
    node.p = v3[0]; node.t = v3[1]; node.l = v3[2];
    emb* nx = update_embedding(e, &node, GEN, cfg);
    embedding_free(e);
    e = nx;
    if (e) ++wk->used;
  }
  *out = e;
  return true;
}

ef_est* ef_get_est_factorizations(const ef_seq* est_info, ef_meg* V, const ef_config* cfg,
                                  const ef_seq* gen_info, ef_backend* be) {
  ef_phase(EFP_EMBED);
  ef_est* est = (ef_est*)calloc(1, sizeof(ef_est));
  est->info = est_info;
  const char* GEN = gen_info->seq;
  const char* EST = est_info->seq;
  const unsigned est_len = (unsigned)V->n - 2;
  ef_list* flist = efl_new();
  ef_work wk = { 0, work_limit(cfg) };
  emb* chain = NULL;
  if (V->rec && !V->v && ef_chain_fast_path && chain_embedding(V->rec, cfg, GEN, &wk, &chain)) {
    if (ef_prof_on) ++ef_prof.chain_graphs;
    if (wk.used > wk.limit) {                          /* budget spent (:190-193): the caller retries */
      if (chain) embedding_free(chain);
      efl_free(flist, ef_factorization_free);
      free(est);
      return NULL;
    }
    if (chain) {
      ef_list* embs = efl_new();
      efl_push_back(embs, chain);
      ef_list* cands = factorizations_from_embeddings(embs, cfg);
      efl_free(embs, embedding_free);
      flist = clean_candidates(cands, flist, est_len, GEN, EST, cfg, be);
    }
  } else {
  if (ef_prof_on) ++ef_prof.other_graphs;
  V = ef_meg_lists(V);                                 /* a graph known by its record only gets its lists now */
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) { ef_pairing* p = (ef_pairing*)efi_next(&it); p->visited = false; p->emb_memo = NULL; }
  }
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) {
      ef_pairing* root = (ef_pairing*)efi_next(&it);
      if (root->visited) continue;
      ef_phase(EFP_EMBED);
      ef_list* embs = subtree_embeddings(root, cfg, GEN, &wk);
      if (!embs) {
        /* budget spent (:190-193): nothing of this attempt is kept, the caller retries */
        EF_MEG_FOR_POS(V, j, 0, V->n) {
          ef_iter jt = efl_begin(V->v[j]);
          while (efi_has_next(&jt)) { ef_pairing* q = (ef_pairing*)efi_next(&jt); if (q->emb_memo) { efl_free(q->emb_memo, embedding_free); q->emb_memo = NULL; } }
        }
        efl_free(flist, ef_factorization_free);
        free(est);
        return NULL;
      }
      flist = clean_candidates(factorizations_from_embeddings(embs, cfg), flist, est_len, GEN, EST, cfg, be);
    }
  }
  /* release the memoised embeddings */
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) { ef_pairing* p = (ef_pairing*)efi_next(&it); if (p->emb_memo) { efl_free(p->emb_memo, embedding_free); p->emb_memo = NULL; } }
  }
  }
  {
    unsigned long long hw = __atomic_load_n(&work_high_water, __ATOMIC_RELAXED);
    while (wk.used > hw && !__atomic_compare_exchange_n(&work_high_water, &hw, wk.used, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
  }

  ef_phase(EFP_FILTERS);
  /* FILTER 1: coverage (:278-331) */
  {
    const size_t nf = efl_size(flist);
    double cov_small[32];
    double* cov = nf < 32 ? cov_small : (double*)malloc((nf + 1) * sizeof(double));
    double max_cov = 0.0;
    size_t k = 0;
    ef_iter it = efl_begin(flist);
    while (efi_has_next(&it)) {
      ef_list* f = (ef_list*)efi_next(&it);
      bool src_sink = false;
      if (efl_size(f) == 1) {
        const ef_factor* h = (const ef_factor*)efl_head(f);
        if (h->EST_start < 0 || h->EST_start >= (int)est_len) { cov[k] = -1.0; src_sink = true; }
      }
      if (!src_sink) { cov[k] = coverage_of(f, est_len); if (max_cov < cov[k]) max_cov = cov[k]; }
      ++k;
    }
    k = 0;
    it = efl_begin(flist);
    const int slen = (int)strlen(EST);
    while (efi_has_next(&it)) {
      efi_next(&it);
      const double c = cov[k++];
      if (c == -1.0 || max_cov - c > cfg->max_coverage_diff) efi_remove(&it, ef_factorization_free);
      else if ((max_cov - c) * slen > 100) efi_remove(&it, ef_factorization_free);
    }
    if (cov != cov_small) free(cov);
  }
  /* FILTER 3: gap length on P (:376-414) */
  {
    const size_t nf = efl_size(flist);
    int gl_small[32];
    int* gl = nf < 32 ? gl_small : (int*)malloc((nf + 1) * sizeof(int));
    int min_gl = -1;
    size_t k = 0;
    ef_iter it = efl_begin(flist);
    while (efi_has_next(&it)) {
      gl[k] = gap_length_of((ef_list*)efi_next(&it));
      if (min_gl == -1 || min_gl > gl[k]) min_gl = gl[k];
      ++k;
    }
    k = 0;
    it = efl_begin(flist);
    while (efi_has_next(&it)) {
      efi_next(&it);
      const int g = gl[k++];
      if (cfg->max_gapLength_diff != -1 && g - min_gl > cfg->max_gapLength_diff) efi_remove(&it, ef_factorization_free);
    }
    if (gl != gl_small) free(gl);
  }
  ef_phase(EFP_GAPERR);
  /* FILTER 4: gap errors (:416-433) */
  if (be->ahead) {
    /* the border refinements of all gaps of all candidates do not depend on each other (a gap reads and writes
     * only the two exon ends that face it): asked in one request, found by the loop below */
    ef_ahead_collect(be, true);
    ef_iter it = efl_begin(flist);
    while (efi_has_next(&it)) check_gap_errors((ef_list*)efi_next(&it), EST, GEN, cfg, be);
    ef_ahead_collect(be, false);
    if (ef_ahead_flush(be) != 0) { fprintf(stderr, "* FATAL dynamic-programming backend failed\n"); abort(); }
  }
  {
    ef_iter it = efl_begin(flist);
    while (efi_has_next(&it)) {
      ef_list* f = (ef_list*)efi_next(&it);
      if (!check_gap_errors(f, EST, GEN, cfg, be)) efi_remove(&it, ef_factorization_free);
    }
  }
  if (cfg->max_number_of_factorizations != 0 && (int)efl_size(flist) > cfg->max_number_of_factorizations) {
    efl_free(flist, ef_factorization_free);
    flist = efl_new();
  }
  ef_phase(EFP_INTRON);
  /* intron refinement (:446-490) */
  {
    /* The gap alignments of the introns are asked together, from the exons as they are now: refining an intron
     * moves the two exon ends that face it, which changes the strings of the NEXT intron only when the exon
     * between them is shorter than the 30-character windows -- every alignment is taken at its turn when its
     * strings are still the same and asked again otherwise (ef_refine_intron). */
    enum { GAP_AHEAD_MAX = 24 };
    ef_gap_ahead* pre = NULL; size_t n_pre = 0;
    if (be->ahead && be->dp_many) {
      size_t total = 0;
      ef_iter it = efl_begin(flist);
      while (efi_has_next(&it)) { const size_t k = efl_size((ef_list*)efi_next(&it)); total += k ? k - 1 : 0; }
      if (total >= 2 && total <= GAP_AHEAD_MAX) {
        pre = (ef_gap_ahead*)malloc(total * sizeof(ef_gap_ahead));
        ef_dp_req rq[GAP_AHEAD_MAX]; ef_dp_res rs[GAP_AHEAD_MAX];
        it = efl_begin(flist);
        while (efi_has_next(&it)) {
          ef_list* f = (ef_list*)efi_next(&it);
          if (efl_empty(f)) continue;
          ef_iter fi = efl_begin(f);
          const ef_factor* donor = (const ef_factor*)efi_next(&fi);
          while (efi_has_next(&fi)) {
            const ef_factor* acc = (const ef_factor*)efi_next(&fi);
            ef_gap_window_build(cfg, gen_info, est_info, donor, acc, &pre[n_pre].w);
            pre[n_pre].donor = *donor; pre[n_pre].acceptor = *acc;
            const ef_dp_req q = { EF_DP_GAP, pre[n_pre].w.seq_est, pre[n_pre].w.le, pre[n_pre].w.seq_gen, pre[n_pre].w.lg, 0, 0, 0, 0, 1 };
            rq[n_pre++] = q;
            donor = acc;
          }
        }
        memset(rs, 0, n_pre * sizeof(ef_dp_res));
        if (be->dp_many(be->self, rq, rs, n_pre) != 0) { fprintf(stderr, "* FATAL gap alignment backend failed\n"); abort(); }
        for (size_t k = 0; k < n_pre; ++k) pre[k].res = rs[k];
        if (ef_prof_on) ef_prof.ahead_asked += n_pre;
      }
    }
    size_t at = 0;
    ef_iter it = efl_begin(flist);
    while (efi_has_next(&it)) {
      ef_list* f = (ef_list*)efi_next(&it);
      if (efl_empty(f)) continue;
      ef_iter fi = efl_begin(f);
      ef_factor* donor = (ef_factor*)efi_next(&fi);
      bool first = true;
      while (efi_has_next(&fi)) {
        ef_factor* acc = (ef_factor*)efi_next(&fi);
        ef_refine_intron(cfg, gen_info, est_info, donor, acc, first, be, pre && at < n_pre ? &pre[at] : NULL);
        ++at;
        first = false;
        donor = acc;
      }
      fi = efl_begin(f);
      const ef_factor* e1 = (const ef_factor*)efi_next(&fi);
      if (efi_has_next(&fi)) {
        const ef_factor* e2 = (const ef_factor*)efi_next(&fi);
        if (e1->EST_start == e2->EST_start) free(efl_pop_front(f));
      }
    }
    for (size_t k = 0; k < n_pre; ++k) { ef_gap_window_release(&pre[k].w); ef_dp_res_release(&pre[k].res); }
    free(pre);
  }
  ef_phase(EFP_TAIL);
  /* tail correction and polyA (:573-585) -- on the ORIGINAL EST sequence */
  est->polyA_signals = efl_new();
  est->polyadenil_signals = efl_new();
  {
    ef_iter it = efl_begin(flist);
    while (efi_has_next(&it)) {
      ef_list* f = (ef_list*)efi_next(&it);
      correct_tail(f, GEN, est_info->original_seq);
      bool polyadenil = false;
      const bool polyA = detect_polyA(f, GEN, est_info->original_seq, &polyadenil);
      efl_push_back(est->polyA_signals, polyA ? (void*)1 : (void*)2);
      efl_push_back(est->polyadenil_signals, polyadenil ? (void*)1 : (void*)2);
    }
  }
  est->factorizations = flist;
  return est;
}
