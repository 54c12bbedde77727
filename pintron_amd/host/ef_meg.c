/* Maximal-embedding graph: edges, simplification, transitive reduction, short-edge compaction,
 * complexity tests, writers.  Behaviour follows src/max-emb-graph.c:394-700,
 * src/meg-simplification.c and src/io-meg.c:146-190 of the reference; list iteration order is
 * part of that behaviour (ef_list.h). */
#include <stdlib.h>
#include <string.h>

#include "estfact.h"

typedef char ef_pairing_fits_cell64[sizeof(ef_pairing) <= sizeof(ef_cell64) ? 1 : -1];
static ef_pairing* pairing_new(int p, int t, int l) {
  ef_pairing* x = (ef_pairing*)ef_cell64_get();
  memset(x, 0, sizeof(ef_pairing));
  x->p = p; x->t = t; x->l = l;
  x->adjs = efl_new(); x->incs = efl_new();
  return x;
}

static void pairing_free(void* v) {
  ef_pairing* x = (ef_pairing*)v;
  efl_free(x->adjs, NULL); efl_free(x->incs, NULL);
  ef_cell64_put(x);
}

/* positions without a vertex share this list: it is never written (vertices are only added at
 * positions that already hold one) */
static ef_list empty_position = { { &empty_position.sent, &empty_position.sent, NULL }, 0 };

ef_meg* ef_meg_from_pairings(const ef_triple* tr, size_t n_tr, size_t m) {
  ef_meg* V = (ef_meg*)malloc(sizeof(ef_meg));
  V->n = m + 2;
  V->rec = NULL; V->slab = false; V->lists = NULL;
  /* one list per EST position, most of them empty: headers exist only for the positions that
   * hold a vertex (source, sink and the distinct p of the pairings) */
  V->v = (ef_list**)malloc(V->n * sizeof(ef_list*) + (n_tr + 2) * sizeof(ef_list));
  ef_list* heads = (ef_list*)(V->v + V->n);
  for (size_t i = 0; i < V->n; ++i) V->v[i] = &empty_position;
  V->act = (size_t*)malloc((n_tr + 2) * sizeof(size_t));
  V->n_act = 0;
  size_t h = 0;
#define MEG_POS(i_) do { if (V->v[(i_)] == &empty_position) { V->v[(i_)] = &heads[h++]; efl_init(V->v[(i_)]); } } while (0)
  MEG_POS(0);
  efl_push_back(V->v[0], pairing_new(EF_SOURCE_START, EF_SOURCE_START, EF_SOURCE_LEN));
  for (size_t k = 0; k < n_tr; ++k) {
    MEG_POS(1 + (size_t)tr[k].p);
    efl_push_back(V->v[1 + tr[k].p], pairing_new(tr[k].p, tr[k].t, tr[k].l));
  }
  MEG_POS(V->n - 1);
  efl_push_back(V->v[V->n - 1], pairing_new(EF_SINK_START, EF_SINK_START, EF_SOURCE_LEN));
#undef MEG_POS
  for (size_t i = 0; i < V->n; ++i) if (V->v[i] != &empty_position) V->act[V->n_act++] = i;
  return V;
}

/* MEG built on the device (libpintron_gpu.so: pgpu_meg.hip) -> the list structure the embedding
 * enumeration and the writers walk.  Vertices come in position-list order with their adjacency
 * lists in the reference's order; incidence lists are not needed past this point and stay empty. */
ef_meg* ef_meg_from_record(const void* rec, size_t m) {
  const uint32_t* head = (const uint32_t*)rec;
  const uint32_t nv = head[0];
  const int32_t* vt = (const int32_t*)((const char*)rec + 16);
  const uint16_t* first = (const uint16_t*)((const char*)rec + 16 + 12 * (size_t)nv);
  const uint8_t* tgt = (const uint8_t*)rec + 16 + 12 * (size_t)nv + 2 * ((size_t)nv + 1);
  const size_t ne = first[nv], n = m + 2;
  /* one block: [ef_meg][v: nv pointers][act: nv][position headers: nv][vertices: nv][adjs + incs
   * headers: 2 nv][nodes: nv (position lists) + ne (adjacency)] -- about a kilobyte for a dozen
   * vertices instead of three pool cells per vertex, one per edge and a table of |P| + 2 pointers, and
   * one free().  What reads such a graph visits "every position that holds vertices, ascending" and
   * never asks which position it is (the vertices carry p), so the position table is compact here:
   * v[k] = the k-th such position's list, act[k] = k (EF_MEG_FOR_POS then yields k). */
  const size_t bytes = sizeof(ef_meg) + nv * sizeof(ef_list*) + nv * sizeof(size_t) + nv * sizeof(ef_list) +
                       nv * sizeof(ef_pairing) + 2 * (size_t)nv * sizeof(ef_list) + ((size_t)nv + ne) * sizeof(ef_node);
  char* blk = (char*)malloc(bytes + 8);
  ef_meg* V = (ef_meg*)blk; blk += sizeof(ef_meg);
  V->n = n; V->rec = rec; V->slab = true; V->lists = NULL;
  V->v = (ef_list**)blk; blk += nv * sizeof(ef_list*);
  V->act = (size_t*)blk; blk += nv * sizeof(size_t);
  ef_list* heads = (ef_list*)blk; blk += nv * sizeof(ef_list);
  ef_pairing* vx = (ef_pairing*)blk; blk += nv * sizeof(ef_pairing);
  ef_list* lists = (ef_list*)blk; blk += 2 * (size_t)nv * sizeof(ef_list);
  ef_node* node = (ef_node*)blk;
  V->n_act = 0;
#define SLAB_PUSH_BACK(l_, el_) do { ef_node* nd_ = node++; nd_->el = (el_); nd_->prev = (l_)->sent.prev; nd_->next = &(l_)->sent; \
                                      (l_)->sent.prev->next = nd_; (l_)->sent.prev = nd_; ++(l_)->size; } while (0)
  size_t last_pos = (size_t)-1;
  for (uint32_t k = 0; k < nv; ++k) {
    const int p = vt[3 * k], t = vt[3 * k + 1], l = vt[3 * k + 2];
    const size_t pos = p == EF_SOURCE_START ? 0 : (p == EF_SINK_START ? n - 1 : 1 + (size_t)p);
    if (pos != last_pos) {                 /* vertices come in position-list order */
      V->v[V->n_act] = &heads[V->n_act]; efl_init(V->v[V->n_act]); V->act[V->n_act] = V->n_act; ++V->n_act;
      last_pos = pos;
    }
    ef_pairing* x = &vx[k];
    x->p = p; x->t = t; x->l = l; x->id = 0; x->visited = false; x->emb_memo = NULL;
    x->adjs = &lists[2 * k]; x->incs = &lists[2 * k + 1];
    efl_init(x->adjs); efl_init(x->incs);
    SLAB_PUSH_BACK(V->v[V->n_act - 1], x);
  }
  for (uint32_t k = 0; k < nv; ++k)
    for (uint32_t e = first[k]; e < first[k + 1]; ++e) SLAB_PUSH_BACK(vx[k].adjs, &vx[tgt[e]]);
#undef SLAB_PUSH_BACK
  return V;
}

ef_meg* ef_meg_record_only(const void* rec, size_t m) {
  ef_meg* V = (ef_meg*)calloc(1, sizeof(ef_meg));
  V->n = m + 2; V->rec = rec; V->slab = true;
  return V;
}

ef_meg* ef_meg_lists(ef_meg* V) {
  if (V->v || !V->rec) return V;
  if (!V->lists) V->lists = ef_meg_from_record(V->rec, V->n - 2);
  return V->lists;
}

void ef_meg_free(ef_meg* V) {
  if (!V) return;
  if (V->lists) ef_meg_free(V->lists);
  if (V->slab) { free(V); return; }
  EF_MEG_FOR_POS(V, i, 0, V->n) efl_clear(V->v[i], pairing_free);
  free(V->v); free(V->act);
  free(V);
}

/* is_there_an_edge_strict (src/max-emb-graph.c:393-465) */
static bool edge_strict(const ef_pairing* I, const ef_pairing* J, int l, int fl, const ef_config* cfg) {
  const double MAX_OVERLAP = 0.4;
  const bool I_is_long = I->l >= 5 * l;
  if (J->p <= I->p) return false;
  if (J->t <= I->t) return false;
  const bool simple_T = (I->t + I->l <= J->t) &&
      (cfg->max_intron_length == 0 || J->t <= I->t + I->l + cfg->max_intron_length);
  const bool overlap_T = (I->t + 2 * l <= J->t + J->l) && (J->t < I->t + I->l) &&
      (J->p + I->t - I->p - J->t <= fl);
  if (I->p + I->l <= J->p && J->p <= I->p + I->l + fl) {          /* simple sequence on P */
    if (simple_T) return true;
    if (overlap_T) {
      if (I_is_long && (I->t + I->l - J->t > MAX_OVERLAP * I->l)) return false;
      return true;
    }
  } else if ((I->p + 2 * l <= J->p + J->l) && (J->p < I->p + I->l)) {   /* overlap on P */
    if (simple_T) return true;
    if (overlap_T) return true;
  }
  return false;
}

/* add_edges_from (src/max-emb-graph.c:533-553): note the bound mixes list indices and
 * pattern coordinates exactly as the reference does */
static void add_edges_from(ef_pairing* I, ef_meg* V, int l, int fl, const ef_config* cfg) {
  const int n = (int)V->n;
  int ubound = I->p + I->l + fl + 1;
  if (n - l < ubound) ubound = n - l;
  if (ubound <= 0) return;
  EF_MEG_FOR_POS(V, j, 0, ubound) {
    ef_iter it = efl_begin(V->v[j]);
    while (efi_has_next(&it)) {
      ef_pairing* J = (ef_pairing*)efi_next(&it);
      if (edge_strict(I, J, l, fl, cfg)) { efl_push_back(I->adjs, J); efl_push_back(J->incs, I); }
    }
  }
}

static bool apart(const ef_pairing* a, const ef_pairing* b) {     /* disjoint on P and on T */
  return ((a->p + a->l <= b->p) || (b->p + b->l <= a->p)) &&
         ((a->t + a->l <= b->t) || (b->t + b->l <= a->t));
}

void ef_build_edge_set(ef_meg* V, const ef_config* cfg) {
  const int L = (int)cfg->min_factor_len;
  const int fl = 2 * L + 1;                                       /* compute_fl */
  EF_MEG_FOR_POS(V, i, 1, V->n - 1) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) add_edges_from((ef_pairing*)efi_next(&it), V, L, fl, cfg);
  }
  const int p_len = (int)V->n - 2;
  /* add_edges_from_source (:555-599) */
  {
    const int max_p = (int)(((double)p_len) * cfg->max_prefix_discarded_rate);
    ef_pairing* source = (ef_pairing*)efl_head(V->v[0]);
    EF_MEG_FOR_POS(V, i, 1, max_p >= 1 ? max_p + 1 : 1) {
      ef_iter it = efl_begin(V->v[i]);
      while (efi_has_next(&it)) {
        ef_pairing* I = (ef_pairing*)efi_next(&it);
        bool possible = true;
        ef_iter in = efl_begin(I->incs);
        while (possible && efi_has_next(&in)) {
          const ef_pairing* inc = (const ef_pairing*)efi_next(&in);
          possible = !apart(inc, I);
          possible = possible && (((inc->p + L) > I->p) || ((inc->t + L) > I->t));
        }
        if (possible) { efl_push_back(source->adjs, I); efl_push_back(I->incs, source); }
      }
    }
  }
  /* add_edges_to_sink (:601-647) */
  {
    const int min_p = (int)(((double)p_len) * (1.0 - cfg->max_suffix_discarded_rate));
    ef_pairing* sink = (ef_pairing*)efl_head(V->v[p_len + 1]);
    EF_MEG_FOR_POS(V, i, 1, p_len >= 1 ? p_len + 1 : 1) {
      ef_iter it = efl_begin(V->v[i]);
      while (efi_has_next(&it)) {
        ef_pairing* I = (ef_pairing*)efi_next(&it);
        if (I->p + I->l < min_p) continue;
        bool possible = true;
        ef_iter ad = efl_begin(I->adjs);
        while (possible && efi_has_next(&ad)) {
          const ef_pairing* adj = (const ef_pairing*)efi_next(&ad);
          possible = !apart(adj, I);
          possible = possible && (((I->p + I->l + L) > (adj->p + adj->l)) || ((I->t + I->l + L) > (adj->t + adj->l)));
        }
        if (possible) { efl_push_back(sink->incs, I); efl_push_back(I->adjs, sink); }
      }
    }
  }
}

void ef_meg_stats(ef_meg* V, size_t* pairings, size_t* edges) {
  *pairings = 0; *edges = 0;
  if (V->rec && !V->v) {                    /* from the record: its vertices, and the end of its CSR offsets */
    const uint32_t nv = ((const uint32_t*)V->rec)[0];
    const uint16_t* first = (const uint16_t*)((const char*)V->rec + 16 + 12 * (size_t)nv);
    *pairings = nv; *edges = first[nv];
    return;
  }
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) { const ef_pairing* p = (const ef_pairing*)efi_next(&it); ++*pairings; *edges += efl_size(p->adjs); }
  }
}

bool ef_is_too_complex_for_compaction(ef_meg* V) {
  size_t tp, te;
  ef_meg_stats(V, &tp, &te);
  return te > 1000 || tp > 2000;
}

bool ef_is_too_complex(ef_meg* V, const ef_config* cfg) {       /* src/meg-simplification.c:89-139 */
  int min_len = 0;
  size_t freq_min_len = 0, tp = 0, te = 0;
  const size_t est_len = V->n - 2;
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) {
      const ef_pairing* p = (const ef_pairing*)efi_next(&it);
      ++tp;
      if (min_len == 0 || p->l < min_len) { min_len = p->l; freq_min_len = 1; }
      else if (p->l == min_len) ++freq_min_len;
      te += efl_size(p->adjs);
    }
  }
  if (tp < 5 || te < 4) return false;
  if (cfg->max_pairings_in_MEG != 0 && tp > cfg->max_pairings_in_MEG &&
      freq_min_len > cfg->max_freq_shortest_pairing * tp)
    return true;
  if (te > 5 * tp || tp > (2 * est_len) / cfg->min_factor_len ||
      (tp > est_len / cfg->min_factor_len && tp >= 50))
    return true;
  return false;
}

/* remove_other_sources_and_sinks (src/meg-simplification.c:142-191) */
static void remove_dangling(ef_meg* V) {
  bool removed;
  do {
    removed = false;
    EF_MEG_FOR_POS(V, i, 1, V->n - 1) {
      ef_iter it = efl_begin(V->v[i]);
      while (efi_has_next(&it)) {
        ef_pairing* I = (ef_pairing*)efi_next(&it);
        if (efl_empty(I->adjs) || efl_empty(I->incs)) {
          removed = true;
          ef_iter a = efl_begin(I->adjs);
          while (efi_has_next(&a)) efl_remove_first(((ef_pairing*)efi_next(&a))->incs, I);
          ef_iter b = efl_begin(I->incs);
          while (efi_has_next(&b)) efl_remove_first(((ef_pairing*)efi_next(&b))->adjs, I);
          efi_remove(&it, pairing_free);
        }
      }
    }
  } while (removed);
}

/* simplify_meg = remove_useless_edges (:193-232) + remove_other_sources_and_sinks */
void ef_simplify_meg(ef_meg* V, const ef_config* cfg) {
  const int g = 2 * (int)cfg->min_factor_len + 3;                /* compute_gl */
  EF_MEG_FOR_POS(V, i, 1, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) {
      ef_pairing* p = (ef_pairing*)efi_next(&it);
      ef_iter a = efl_begin(p->adjs);
      while (efi_has_next(&a)) {
        ef_pairing* x = (ef_pairing*)efi_next(&a);
        if (x->t == EF_SINK_START) continue;
        int gap = x->t - x->p - p->t + p->p;
        if (gap < 0) gap = 0;
        if (gap > g && gap < cfg->min_intron_length) { efi_remove(&a, NULL); efl_remove_first(x->incs, p); }
      }
    }
  }
  remove_dangling(V);
}

/* compact_short_edges (src/meg-simplification.c:258-312) */
void ef_compact_short_edges(ef_meg* V, const ef_config* cfg) {
  (void)cfg;
  bool removed;
  do {
    removed = false;
    EF_MEG_FOR_POS(V, i, 1, V->n) {
      ef_iter it = efl_begin(V->v[i]);
      while (efi_has_next(&it)) {
        ef_pairing* p = (ef_pairing*)efi_next(&it);
        ef_iter a = efl_begin(p->adjs);
        while (efi_has_next(&a)) {
          ef_pairing* x = (ef_pairing*)efi_next(&a);
          if (x->t == EF_SINK_START) continue;
          bool compact = false;
          if (x->t + x->l - p->t == x->p + x->l - p->p)
            compact = (x->t >= p->t + p->l) && (x->t - p->t - p->l <= 3);
          if (!compact) continue;
          removed = true;
          efi_remove(&a, NULL);
          efl_remove_first(x->incs, p);
          ef_pairing* nv = pairing_new(p->p, p->t, x->p + x->l - p->p);
          ef_iter q = efl_begin(x->adjs);                       /* copy_adjacencies(new, a) */
          while (efi_has_next(&q)) { ef_pairing* y = (ef_pairing*)efi_next(&q); efl_push_back(nv->adjs, y); efl_push_back(y->incs, nv); }
          q = efl_begin(p->incs);                               /* copy_incidencies(new, p) */
          while (efi_has_next(&q)) { ef_pairing* y = (ef_pairing*)efi_next(&q); efl_push_back(nv->incs, y); efl_push_back(y->adjs, nv); }
          efl_push_back(V->v[i], nv);
        }
      }
    }
    remove_dangling(V);
  } while (removed);
}

/* meg2graph + topological_sort + transitive_reduction (src/meg-simplification.c:333-632) */
static int cmp_by_id(const void* a, const void* b) {
  return (*(ef_pairing* const*)a)->id - (*(ef_pairing* const*)b)->id;
}

void ef_transitive_reduction(ef_meg* V) {
  size_t nv = 0, dummy;
  ef_meg_stats(V, &nv, &dummy);
  /* one scratch block for every per-vertex array of this call */
  const size_t np1 = nv + 1;
  char* scratch = (char*)malloc(np1 * (5 * sizeof(void*) + 2 * sizeof(int) + 1) + (4 * nv + 16) * sizeof(int));
  ef_pairing** G = (ef_pairing**)scratch;
  ef_pairing** T = G + np1;
  ef_list** star = (ef_list**)(T + np1);
  ef_list** red = star + np1;
  ef_list** red_inc = red + np1;
  int* color = (int*)(red_inc + np1);
  int* ids = color + np1;
  int* stack0 = ids + np1;
  unsigned char* in_star = (unsigned char*)(stack0 + 4 * nv + 16);
  size_t k = 0;
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) { G[k] = (ef_pairing*)efi_next(&it); G[k]->id = (int)k; ++k; }
  }
  /* dfs_visit (:358-463): explicit stack, sources first, finishing order gives the ids */
  memset(color, 0, np1 * sizeof(int));
  int* stack = stack0;
  size_t scap = 4 * nv + 16, sp = 0;
  bool acyclic = true;
#define PUSH(x) do { if (sp == scap) { int* bigger = (int*)malloc(2 * scap * sizeof(int)); memcpy(bigger, stack, scap * sizeof(int)); \
                       if (stack != stack0) { free(stack); } \
                       stack = bigger; scap *= 2; } stack[sp++] = (x); } while (0)
  for (size_t i = 0; i < nv; ++i) if (efl_size(G[i]->incs) == 0) PUSH((int)i);
  if (sp == 0) acyclic = false;
  size_t progr = nv;
  do {
    while (sp > 0) {
      const int v = stack[--sp];
      if (color[v] == 0) {
        color[v] = 1;
        PUSH(v);
        ef_iter a = efl_begin(G[v]->adjs);
        while (efi_has_next(&a)) {
          const ef_pairing* w = (const ef_pairing*)efi_next(&a);
          if (color[w->id] == 0) PUSH(w->id);
          else if (color[w->id] == 1) acyclic = false;
        }
      } else if (color[v] == 1) {
        color[v] = 2;
        ids[v] = (int)--progr;
      }
    }
    for (size_t i = 0; i < nv && sp == 0; ++i)
      if (color[i] == 0) { acyclic = false; PUSH((int)i); }
  } while (sp > 0);
#undef PUSH
  if (stack != stack0) free(stack);
  if (!acyclic) {
    fprintf(stderr, "* FATAL The graph is cyclic. Transitive reduction not possible! Terminating.\n");
    abort();
  }
  /* topological order = array order; adjacency lists sorted by id (:465-516) */
  for (size_t i = 0; i < nv; ++i) { G[i]->id = ids[i]; T[ids[i]] = G[i]; }
  for (size_t i = 0; i < nv; ++i) { efl_sort(T[i]->adjs, cmp_by_id); efl_sort(T[i]->incs, cmp_by_id); }
  /* reduction (:518-632) */
  for (size_t i = 0; i < nv; ++i) { star[i] = efl_new(); red[i] = efl_new(); red_inc[i] = efl_new(); }
  for (size_t i = nv; i-- > 0;) {
    ef_pairing* v = T[i];
    memset(in_star, 0, nv);
    in_star[i] = 1;
    efl_push_back(star[i], v);
    ef_iter a = efl_begin(v->adjs);
    while (efi_has_next(&a)) {
      ef_pairing* w = (ef_pairing*)efi_next(&a);
      const bool ends_earlier = (w->p + w->l < v->p + v->l) || (w->t + w->l < v->t + v->l);
      if (!in_star[w->id] || (w->p < v->p) || (w->t < v->t) || ends_earlier) {
        efl_push_back(red[i], w);
        efl_push_back(red_inc[w->id], v);
        if (!ends_earlier) {
          ef_iter s = efl_begin(star[w->id]);
          while (efi_has_next(&s)) {
            ef_pairing* wa = (ef_pairing*)efi_next(&s);
            if (in_star[wa->id]) continue;
            if ((v->t <= wa->t) && (v->p <= wa->p) && (v->t + v->l <= wa->t + wa->l) && (v->p + v->l <= wa->p + wa->l)) {
              in_star[wa->id] = 1;
              efl_push_back(star[i], wa);
            }
          }
        }
      }
    }
  }
  for (size_t i = 0; i < nv; ++i) {
    efl_free(star[i], NULL);
    efl_free(T[i]->adjs, NULL); efl_free(T[i]->incs, NULL);
    T[i]->adjs = red[i]; T[i]->incs = red_inc[i];
  }
  free(scratch);
}

/* the two texts of a device-built record (layout: include/pintron_gpu.h) */
static const char* record_texts(const void* rec, uint32_t* meg_len, uint32_t* edges_len) {
  const uint32_t* head = (const uint32_t*)rec;
  const size_t graph = (16 + 12 * (size_t)head[0] + 2 * ((size_t)head[0] + 1) + head[1] + 3) & ~(size_t)3;
  const uint32_t* tl = (const uint32_t*)((const char*)rec + graph);
  *meg_len = tl[0]; *edges_len = tl[1];
  return (const char*)rec + graph + 8;
}

void ef_meg_write(ef_sink* f, ef_meg* V) {
  if (V->rec) {
    uint32_t ml, el;
    const char* t = record_texts(V->rec, &ml, &el);
    ef_sink_write(f, t, ml);
    return;
  }
  ef_wbuf w; efw_open(&w, f);
  int index = 0;
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) {
      ef_pairing* p = (ef_pairing*)efi_next(&it);
      efw_ch(&w, '('); efw_int(&w, p->p); efw_ch(&w, ','); efw_int(&w, p->t); efw_ch(&w, ','); efw_int(&w, p->l);
      efw_mem(&w, ")\n", 2);                                   /* "(%d,%d,%d)\n" */
      p->id = index++;
    }
  }
  efw_mem(&w, "#adj#\n", 6);
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) {
      const ef_pairing* p = (const ef_pairing*)efi_next(&it);
      ef_iter a = efl_begin(p->adjs);
      while (efi_has_next(&a)) {                               /* "%d-%d\n" */
        efw_int(&w, p->id); efw_ch(&w, '-'); efw_int(&w, ((const ef_pairing*)efi_next(&a))->id); efw_ch(&w, '\n');
      }
    }
  }
  efw_flush(&w);
}

void ef_intronic_edges_write(ef_sink* f, ef_meg* V) {
  if (V->rec) {
    uint32_t ml, el;
    const char* t = record_texts(V->rec, &ml, &el);
    ef_sink_write(f, t + ml, el);
    return;
  }
  ef_wbuf w; efw_open(&w, f);
  EF_MEG_FOR_POS(V, i, 0, V->n) {
    ef_iter it = efl_begin(V->v[i]);
    while (efi_has_next(&it)) {
      const ef_pairing* p = (const ef_pairing*)efi_next(&it);
      if (p->p == EF_SOURCE_START || p->p == EF_SINK_START) continue;
      ef_iter a = efl_begin(p->adjs);
      while (efi_has_next(&a)) {
        const ef_pairing* x = (const ef_pairing*)efi_next(&a);
        if (x->p == EF_SINK_START) continue;
        const int v9[9] = { p->t + p->l, x->t, p->p + p->l, x->p, (x->t - p->t - p->l), (x->p - p->p - p->l),
                            (x->t - p->t) - (x->p - p->p), p->l, x->l };
        for (int k = 0; k < 9; ++k) { if (k) efw_ch(&w, ' '); efw_int(&w, v9[k]); }
        if ((x->t - p->t) - (x->p - p->p) >= 50) efw_mem(&w, " intronic", 9);
        efw_ch(&w, '\n');
      }
    }
  }
  efw_flush(&w);
}

/* build_meg (src/compute-est-fact.c:90-152) */
ef_meg* ef_build_meg(const ef_seq* est, ef_backend* be, const ef_config* shared, size_t* inc) {
  ef_config cfg = *shared;
  const size_t m = strlen(est->seq);
  ef_meg* V = NULL;
  bool too_complex;
  do {
    cfg.min_factor_len += (unsigned)*inc;
    /* the device may already hold the finished graph of this pattern at these parameters (built
     * right behind the pairings); bit 0 of its flags is build_meg's too_complex, bit 1 says the
     * graph was beyond the device's limits and has to be built here */
    const uint32_t* rec = be->meg ? (const uint32_t*)be->meg(be->self, est->seq, m, &cfg) : NULL;
    if (rec && !(rec[2] & 2u)) {
      too_complex = (rec[2] & 1u) != 0;
      cfg.min_factor_len -= (unsigned)*inc;
      if (too_complex && cfg.min_factor_len + *inc + 1 + 2 < m + 2) { ++*inc; continue; }
      too_complex = false;
      V = ef_meg_record_only(rec, m);
      continue;
    }
    ef_triple* tr = NULL; size_t ntr = 0;
    if (be->pairings(be->self, est->seq, m, cfg.min_factor_len, cfg.min_string_depth_rate, &tr, &ntr) != 0) {
      fprintf(stderr, "* FATAL pairing backend failed\n");
      abort();
    }
    V = ef_meg_from_pairings(tr, ntr, m);
    free(tr);
    ef_build_edge_set(V, &cfg);
    ef_simplify_meg(V, &cfg);
    if (cfg.trans_red) ef_transitive_reduction(V);
    too_complex = ef_is_too_complex_for_compaction(V);
    if (!too_complex && cfg.short_edge_comp) ef_compact_short_edges(V, &cfg);
    too_complex = too_complex || ef_is_too_complex(V, &cfg);
    cfg.min_factor_len -= (unsigned)*inc;
    if (too_complex) {
      if (cfg.min_factor_len + *inc + 1 + 2 < V->n) {
        ++*inc;
        ef_meg_free(V);
        V = NULL;
      } else {
        too_complex = false;
      }
    }
  } while (too_complex);
  return V;
}
