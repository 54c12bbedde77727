// Maximal-embedding graph of every pattern of a pairing plan, built on the device right behind the
// pairing kernels (the pairings never leave HBM):
//   build_edge_set            src/max-emb-graph.c:650-676 (is_there_an_edge_strict :394-465,
//                             add_edges_from :533-553, add_edges_from_source :555-599,
//                             add_edges_to_sink :601-647)
//   simplify_meg              src/meg-simplification.c:314 (remove_useless_edges :193-232,
//                             remove_other_sources_and_sinks :142-191)
//   transitive_reduction      src/meg-simplification.c:333-632 (meg2graph, dfs_visit, topological_sort)
//   compact_short_edges       src/meg-simplification.c:258-312
//   is_too_complex(_for_compaction)   src/meg-simplification.c:68-139
// i.e. the body of build_meg (src/compute-est-fact.c:101-131) after build_vertex_set.
//
// The reference's result depends on the order of its linked lists (position lists, adjacency and
// incidence lists) and on its iterator, which caches the NEXT node when it hands out an element.
// Here a MEG is a set of small ordered ARRAYS in a per-pattern scratch block; one thread builds one
// MEG (the graphs have a dozen vertices; 25 000 of them per launch).  The array operations below
// keep the list semantics:
//   order[]   all live vertices sorted by (position list, place in the list); a vertex appended to
//             the list of position i goes behind the last vertex of position i;
//   adj/inc   per-vertex ordered arrays (push_back, remove-first-equal, remove-at-cursor, sort).
// A MEG that does not fit the caps below (or has a cycle, which the reference treats as fatal) is
// flagged PGPU_MEG_UNAVAILABLE and the caller builds it itself from the pairings.
#include "pgpu_index.h"

namespace {

constexpr int MV = PGPU_MEG_MAX_VERTICES;    // vertices ever created per MEG (source, sink, pairings, compactions)
constexpr int MD = PGPU_MEG_MAX_DEGREE;      // out- or in-degree of a vertex
constexpr int MS = 1024;                     // DFS stack entries
constexpr int32_t SRC_START = INT32_MIN, SINK_START = INT32_MAX - 200, SRC_LEN = 200;   // include/types.h:203-206

struct MegScratch {
  int32_t p[MV], t[MV], l[MV];
  uint32_t pos[MV];                 // index of the position list the vertex lives in: 0 source, 1+p, n-1 sink
  uint8_t order[MV];
  uint8_t nadj[MV], ninc[MV];
  uint8_t adj[MV][MD], inc[MV][MD];
  uint8_t ids[MV], tv[MV], color[MV], idx_of[MV];
  unsigned long long star[MV];
  uint8_t nred[MV], nrinc[MV];
  uint8_t red[MV][MD], rinc[MV][MD];
  uint8_t stack[MS];
};

struct Meg {
  MegScratch* S;
  int nv;          // vertices created
  int no;          // live vertices (entries of order[])
  int n;           // |P| + 2
  bool overflow;

  __device__ int new_vertex(int32_t p, int32_t t, int32_t l, uint32_t pos) {
    if (nv >= MV) { overflow = true; return MV - 1; }
    const int v = nv++;
    S->p[v] = p; S->t[v] = t; S->l[v] = l; S->pos[v] = pos; S->nadj[v] = 0; S->ninc[v] = 0;
    return v;
  }
  __device__ void push(uint8_t* list, uint8_t& cnt, int v) {
    if (cnt >= MD) { overflow = true; return; }
    list[cnt++] = (uint8_t)v;
  }
  __device__ static void remove_at(uint8_t* list, uint8_t& cnt, int at) {
    for (int k = at; k + 1 < cnt; ++k) list[k] = list[k + 1];
    --cnt;
  }
  __device__ static void remove_first(uint8_t* list, uint8_t& cnt, int v) {
    for (int k = 0; k < cnt; ++k) if (list[k] == v) { remove_at(list, cnt, k); return; }
  }
  __device__ void add_edge(int from, int to) { push(S->adj[from], S->nadj[from], to); push(S->inc[to], S->ninc[to], from); }
  // append v to the position list it belongs to: behind the last live vertex of that position
  __device__ void order_insert(int v) {
    int at = no;
    while (at > 0 && S->pos[S->order[at - 1]] > S->pos[v]) --at;
    for (int k = no; k > at; --k) S->order[k] = S->order[k - 1];
    S->order[at] = (uint8_t)v;
    ++no;
  }
  __device__ void order_remove_at(int at) {
    for (int k = at; k + 1 < no; ++k) S->order[k] = S->order[k + 1];
    --no;
  }

  // is_there_an_edge_strict (src/max-emb-graph.c:394-465)
  __device__ bool edge_strict(int I, int J, int L, int fl, int max_intron) const {
    const int32_t Ip = S->p[I], It = S->t[I], Il = S->l[I], Jp = S->p[J], Jt = S->t[J], Jl = S->l[J];
    if (Jp <= Ip) return false;
    if (Jt <= It) return false;
    const bool long_I = Il >= 5 * L;
    const bool simple_T = (It + Il <= Jt) && (max_intron == 0 || Jt <= It + Il + max_intron);
    const bool overlap_T = (It + 2 * L <= Jt + Jl) && (Jt < It + Il) && (Jp + It - Ip - Jt <= fl);
    if (Ip + Il <= Jp && Jp <= Ip + Il + fl) {
      if (simple_T) return true;
      if (overlap_T) {
        if (long_I && ((double)(It + Il - Jt) > 0.4 * (double)Il)) return false;     // MAX_OVERLAP
        return true;
      }
    } else if ((Ip + 2 * L <= Jp + Jl) && (Jp < Ip + Il)) {
      if (simple_T) return true;
      if (overlap_T) return true;
    }
    return false;
  }
  __device__ bool apart(int a, int b) const {
    return ((S->p[a] + S->l[a] <= S->p[b]) || (S->p[b] + S->l[b] <= S->p[a])) &&
           ((S->t[a] + S->l[a] <= S->t[b]) || (S->t[b] + S->l[b] <= S->t[a]));
  }

  // build_edge_set (src/max-emb-graph.c:650-676)
  __device__ void build_edges(const pgpu_meg_params& prm) {
    const int L = (int)prm.min_factor_len, fl = 2 * L + 1;
    for (int k = 0; k < no; ++k) {
      const int I = S->order[k];
      if (S->pos[I] < 1 || S->pos[I] >= (uint32_t)(n - 1)) continue;
      int ubound = S->p[I] + S->l[I] + fl + 1;                // add_edges_from (:533-553)
      if (n - L < ubound) ubound = n - L;
      if (ubound <= 0) continue;
      for (int kk = 0; kk < no && S->pos[S->order[kk]] < (uint32_t)ubound; ++kk) {
        const int J = S->order[kk];
        if (edge_strict(I, J, L, fl, prm.max_intron_length)) add_edge(I, J);
      }
    }
    const int p_len = n - 2;
    const int source = S->order[0], sink = S->order[no - 1];
    {                                                        // add_edges_from_source (:555-599)
      const int max_p = (int)(((double)p_len) * prm.max_prefix_discarded_rate);
      const uint32_t hi = max_p >= 1 ? (uint32_t)max_p + 1 : 1;
      for (int k = 0; k < no; ++k) {
        const int I = S->order[k];
        if (S->pos[I] < 1 || S->pos[I] >= hi) continue;
        bool possible = true;
        for (int q = 0; possible && q < S->ninc[I]; ++q) {
          const int in = S->inc[I][q];
          possible = !apart(in, I);
          possible = possible && ((S->p[in] + L > S->p[I]) || (S->t[in] + L > S->t[I]));
        }
        if (possible) add_edge(source, I);
      }
    }
    {                                                        // add_edges_to_sink (:601-647)
      const int min_p = (int)(((double)p_len) * (1.0 - prm.max_suffix_discarded_rate));
      const uint32_t hi = p_len >= 1 ? (uint32_t)p_len + 1 : 1;
      for (int k = 0; k < no; ++k) {
        const int I = S->order[k];
        if (S->pos[I] < 1 || S->pos[I] >= hi) continue;
        if (S->p[I] + S->l[I] < min_p) continue;
        bool possible = true;
        for (int q = 0; possible && q < S->nadj[I]; ++q) {
          const int a = S->adj[I][q];
          possible = !apart(a, I);
          possible = possible && ((S->p[I] + S->l[I] + L > S->p[a] + S->l[a]) || (S->t[I] + S->l[I] + L > S->t[a] + S->l[a]));
        }
        if (possible) add_edge(I, sink);
      }
    }
  }

  // remove_other_sources_and_sinks (src/meg-simplification.c:142-191)
  __device__ void remove_dangling() {
    bool removed;
    do {
      removed = false;
      for (int k = 0; k < no;) {
        const int I = S->order[k];
        if (S->pos[I] < 1 || S->pos[I] >= (uint32_t)(n - 1)) { ++k; continue; }
        if (S->nadj[I] == 0 || S->ninc[I] == 0) {
          removed = true;
          for (int q = 0; q < S->nadj[I]; ++q) { const int a = S->adj[I][q]; remove_first(S->inc[a], S->ninc[a], I); }
          for (int q = 0; q < S->ninc[I]; ++q) { const int b = S->inc[I][q]; remove_first(S->adj[b], S->nadj[b], I); }
          S->nadj[I] = 0; S->ninc[I] = 0;
          order_remove_at(k);                                 // the next vertex moves to k
        } else ++k;
      }
    } while (removed);
  }

  // simplify_meg = remove_useless_edges (:193-232) + remove_other_sources_and_sinks
  __device__ void simplify(const pgpu_meg_params& prm) {
    const int g = 2 * (int)prm.min_factor_len + 3;                       // compute_gl
    for (int k = 0; k < no; ++k) {
      const int pv = S->order[k];
      if (S->pos[pv] < 1) continue;
      for (int a = 0; a < S->nadj[pv];) {
        const int x = S->adj[pv][a];
        if (S->t[x] == SINK_START) { ++a; continue; }
        int gap = S->t[x] - S->p[x] - S->t[pv] + S->p[pv];
        if (gap < 0) gap = 0;
        if (gap > g && gap < prm.min_intron_length) { remove_at(S->adj[pv], S->nadj[pv], a); remove_first(S->inc[x], S->ninc[x], pv); }
        else ++a;
      }
    }
    remove_dangling();
  }

  __device__ void sort_by_id(uint8_t* list, int cnt) const {          // ids are distinct
    for (int a = 1; a < cnt; ++a) {
      const uint8_t key = list[a];
      int b = a;
      while (b > 0 && S->ids[S->idx_of[list[b - 1]]] > S->ids[S->idx_of[key]]) { list[b] = list[b - 1]; --b; }
      list[b] = key;
    }
  }

  // meg2graph + dfs_visit + topological_sort + transitive_reduction (src/meg-simplification.c:333-632).
  // false: the graph has a cycle (fatal in the reference) or the DFS stack does not fit.
  __device__ bool transitive_reduction() {
    const int nvx = no;
    // G[k] = order[k]; idx_of[vertex] = k
    for (int k = 0; k < nvx; ++k) { S->idx_of[S->order[k]] = (uint8_t)k; S->color[k] = 0; }
    int sp = 0;
    bool acyclic = true, fits = true;
#define MEG_PUSH(x) do { if (sp >= MS) { fits = false; } else S->stack[sp++] = (uint8_t)(x); } while (0)
    for (int k = 0; k < nvx; ++k) if (S->ninc[S->order[k]] == 0) MEG_PUSH(k);
    if (sp == 0) acyclic = false;
    int progr = nvx;
    do {
      while (sp > 0 && fits) {
        const int v = S->stack[--sp];
        if (S->color[v] == 0) {
          S->color[v] = 1;
          MEG_PUSH(v);
          const int vv = S->order[v];
          for (int q = 0; q < S->nadj[vv]; ++q) {
            const int w = S->idx_of[S->adj[vv][q]];
            if (S->color[w] == 0) MEG_PUSH(w);
            else if (S->color[w] == 1) acyclic = false;
          }
        } else if (S->color[v] == 1) {
          S->color[v] = 2;
          S->ids[v] = (uint8_t)--progr;
        }
      }
      if (!fits) return false;
      for (int k = 0; k < nvx && sp == 0; ++k)
        if (S->color[k] == 0) { acyclic = false; MEG_PUSH(k); }
    } while (sp > 0);
#undef MEG_PUSH
    if (!acyclic) return false;
    // topological order: tv[id] = vertex; adjacency and incidence lists sorted by id (:465-516)
    for (int k = 0; k < nvx; ++k) S->tv[S->ids[k]] = S->order[k];
    for (int k = 0; k < nvx; ++k) { const int v = S->order[k]; sort_by_id(S->adj[v], S->nadj[v]); sort_by_id(S->inc[v], S->ninc[v]); }
    for (int i = 0; i < nvx; ++i) { S->nred[i] = 0; S->nrinc[i] = 0; }
    // reduction (:518-632); star[i] == the in_star marks of round i, as a bit set over ids
    for (int i = nvx; i-- > 0;) {
      const int v = S->tv[i];
      unsigned long long st = 1ull << i;
      const int32_t vp = S->p[v], vt = S->t[v], vl = S->l[v];
      for (int q = 0; q < S->nadj[v]; ++q) {
        const int w = S->adj[v][q];
        const int wid = S->ids[S->idx_of[w]];
        const bool ends_earlier = (S->p[w] + S->l[w] < vp + vl) || (S->t[w] + S->l[w] < vt + vl);
        if (!((st >> wid) & 1ull) || (S->p[w] < vp) || (S->t[w] < vt) || ends_earlier) {
          push(S->red[i], S->nred[i], w);
          push(S->rinc[wid], S->nrinc[wid], v);
          if (!ends_earlier) {
            unsigned long long cand = S->star[wid] & ~st;
            while (cand) {
              const int b = __builtin_ctzll(cand);
              cand &= cand - 1;
              const int wa = S->tv[b];
              if ((vt <= S->t[wa]) && (vp <= S->p[wa]) && (vt + vl <= S->t[wa] + S->l[wa]) && (vp + vl <= S->p[wa] + S->l[wa]))
                st |= 1ull << b;
            }
          }
        }
      }
      S->star[i] = st;
    }
    for (int i = 0; i < nvx; ++i) {
      const int v = S->tv[i];
      S->nadj[v] = S->nred[i]; S->ninc[v] = S->nrinc[i];
      for (int q = 0; q < S->nred[i]; ++q) S->adj[v][q] = S->red[i][q];
      for (int q = 0; q < S->nrinc[i]; ++q) S->inc[v][q] = S->rinc[i][q];
    }
    return true;
  }

  __device__ void stats(uint32_t* tp, uint32_t* te) const {
    uint32_t e = 0;
    for (int k = 0; k < no; ++k) e += S->nadj[S->order[k]];
    *tp = (uint32_t)no; *te = e;
  }

  // compact_short_edges (src/meg-simplification.c:258-312)
  __device__ void compact_short_edges() {
    bool removed;
    do {
      removed = false;
      for (int k = 0; k < no;) {
        const int pv = S->order[k];
        const uint32_t i = S->pos[pv];
        if (i < 1) { ++k; continue; }
        // the reference's iterator has already picked its next node when it hands pv out: a vertex
        // appended to this position's list is visited in this pass unless pv was the last one
        const bool was_last = (k + 1 == no) || (S->pos[S->order[k + 1]] != i);
        for (int a = 0; a < S->nadj[pv];) {
          const int x = S->adj[pv][a];
          if (S->t[x] == SINK_START) { ++a; continue; }
          bool compact = false;
          if (S->t[x] + S->l[x] - S->t[pv] == S->p[x] + S->l[x] - S->p[pv])
            compact = (S->t[x] >= S->t[pv] + S->l[pv]) && (S->t[x] - S->t[pv] - S->l[pv] <= 3);
          if (!compact) { ++a; continue; }
          removed = true;
          remove_at(S->adj[pv], S->nadj[pv], a);
          remove_first(S->inc[x], S->ninc[x], pv);
          const int nvx = new_vertex(S->p[pv], S->t[pv], S->p[x] + S->l[x] - S->p[pv], i);
          if (overflow) return;
          for (int q = 0; q < S->nadj[x]; ++q) { const int y = S->adj[x][q]; add_edge(nvx, y); }              // copy_adjacencies
          for (int q = 0; q < S->ninc[pv]; ++q) { const int y = S->inc[pv][q]; push(S->inc[nvx], S->ninc[nvx], y); push(S->adj[y], S->nadj[y], nvx); }   // copy_incidencies
          order_insert(nvx);
          if (overflow) return;
        }
        if (was_last) { ++k; while (k < no && S->pos[S->order[k]] == i) ++k; }
        else ++k;
      }
      remove_dangling();
    } while (removed);
  }

  // is_too_complex (src/meg-simplification.c:89-139)
  __device__ bool too_complex(const pgpu_meg_params& prm) const {
    int32_t min_len = 0;
    uint32_t freq = 0, tp = 0, te = 0;
    const uint32_t est_len = (uint32_t)(n - 2);
    for (int k = 0; k < no; ++k) {
      const int v = S->order[k];
      ++tp;
      if (min_len == 0 || S->l[v] < min_len) { min_len = S->l[v]; freq = 1; }
      else if (S->l[v] == min_len) ++freq;
      te += S->nadj[v];
    }
    if (tp < 5 || te < 4) return false;
    if (prm.max_pairings_in_MEG != 0 && tp > prm.max_pairings_in_MEG && (double)freq > prm.max_freq_shortest_pairing * (double)tp) return true;
    if (te > 5 * tp || tp > (2 * est_len) / prm.min_factor_len || (tp > est_len / prm.min_factor_len && tp >= 50)) return true;
    return false;
  }
};

// ---- the text est-fact prints for a MEG (the device formats what it has built) -----------------
__device__ __forceinline__ uint32_t dec_len(int32_t v) {               // strlen of printf("%d", v)
  uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v, n = v < 0 ? 2u : 1u;
  while (u >= 10u) { u /= 10u; ++n; }
  return n;
}
__device__ __forceinline__ uint8_t* put_dec(uint8_t* w, int32_t v) {
  uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
  if (v < 0) *w++ = '-';
  uint8_t tmp[10]; int k = 0;
  do { tmp[k++] = (uint8_t)('0' + u % 10u); u /= 10u; } while (u);
  while (k) *w++ = tmp[--k];
  return w;
}
// the nine numbers of a meg-edges.txt line (add_intronic_edges_to_file, src/max-emb-graph.c:677-699)
__device__ __forceinline__ void edge_fields(const MegScratch* S, int pv, int x, int32_t* v9) {
  v9[0] = S->t[pv] + S->l[pv]; v9[1] = S->t[x]; v9[2] = S->p[pv] + S->l[pv]; v9[3] = S->p[x];
  v9[4] = S->t[x] - S->t[pv] - S->l[pv]; v9[5] = S->p[x] - S->p[pv] - S->l[pv];
  v9[6] = (S->t[x] - S->t[pv]) - (S->p[x] - S->p[pv]); v9[7] = S->l[pv]; v9[8] = S->l[x];
}

// one thread = one pattern.  info[pat] = {n_vertices, n_edges, flags, live vertices}
__global__ __launch_bounds__(64)
void meg_build_kernel(const pgpu_pairing* __restrict__ pairs, const unsigned long long* __restrict__ pair_first,
                      const unsigned long long* __restrict__ pat_off, uint32_t n_pat, pgpu_meg_params prm,
                      MegScratch* __restrict__ scratch, uint4* __restrict__ info, uint32_t* __restrict__ rec_bytes) {
  const uint32_t pat = blockIdx.x * blockDim.x + threadIdx.x;
  if (pat >= n_pat) return;
  const uint32_t m = (uint32_t)(pat_off[pat + 1] - pat_off[pat]);
  const unsigned long long f0 = pair_first[pat], f1 = pair_first[pat + 1];
  Meg M;
  M.S = scratch + pat; M.nv = 0; M.no = 0; M.n = (int)m + 2; M.overflow = false;
  uint32_t flags = 0;
  if (f1 - f0 + 2 > (unsigned long long)MV) M.overflow = true;
  else {
    // ef_meg_from_pairings: source, the pairings in list order, sink
    M.S->order[M.no++] = (uint8_t)M.new_vertex(SRC_START, SRC_START, SRC_LEN, 0);
    for (unsigned long long k = f0; k < f1; ++k) {
      const pgpu_pairing q = pairs[k];
      M.S->order[M.no++] = (uint8_t)M.new_vertex(q.p, q.t, q.l, 1u + (uint32_t)q.p);
    }
    M.S->order[M.no++] = (uint8_t)M.new_vertex(SINK_START, SINK_START, SRC_LEN, m + 1);
    M.build_edges(prm);
    if (!M.overflow) M.simplify(prm);
    if (!M.overflow && prm.trans_red && !M.transitive_reduction()) M.overflow = true;
    if (!M.overflow) {
      uint32_t tp, te;
      M.stats(&tp, &te);
      bool complex = te > 1000 || tp > 2000;                 // is_too_complex_for_compaction (:68-87)
      if (!complex && prm.short_edge_comp) M.compact_short_edges();
      if (!M.overflow) complex = complex || M.too_complex(prm);
      if (complex) flags |= PGPU_MEG_TOO_COMPLEX;
    }
  }
  if (M.overflow) {
    info[pat] = make_uint4(0, 0, PGPU_MEG_UNAVAILABLE, 0);
    rec_bytes[pat] = 16;
    return;
  }
  uint32_t tp, te;
  M.stats(&tp, &te);
  // final numbering = position-list order (what meg_write prints, src/io-meg.c:161-170)
  for (int k = 0; k < M.no; ++k) M.S->idx_of[M.S->order[k]] = (uint8_t)k;
  info[pat] = make_uint4(tp, te, flags, (uint32_t)M.no);
  // text lengths: meg_write (src/io-meg.c:146-190) and add_intronic_edges_to_file
  uint32_t meg_txt = 6, edge_txt = 0;                                   // "#adj#\n"
  for (int k = 0; k < M.no; ++k) {
    const int v = M.S->order[k];
    meg_txt += 5 + dec_len(M.S->p[v]) + dec_len(M.S->t[v]) + dec_len(M.S->l[v]);      // "(%d,%d,%d)\n"
    const bool inner = M.S->p[v] != SRC_START && M.S->p[v] != SINK_START;
    for (int q = 0; q < M.S->nadj[v]; ++q) {
      const int x = M.S->adj[v][q];
      meg_txt += 2 + dec_len(k) + dec_len((int32_t)M.S->idx_of[x]);                    // "%d-%d\n"
      if (inner && M.S->p[x] != SINK_START) {
        int32_t v9[9];
        edge_fields(M.S, v, x, v9);
        edge_txt += 9;                                                                 // 8 blanks + newline
        for (int f = 0; f < 9; ++f) edge_txt += dec_len(v9[f]);
        if (v9[6] >= 50) edge_txt += 9;                                                // " intronic"
      }
    }
  }
  M.S->star[0] = ((unsigned long long)edge_txt << 32) | meg_txt;         // handed to the emit kernel
  const uint32_t graph = (16 + 12 * tp + 2 * (tp + 1) + te + 3u) & ~3u;
  rec_bytes[pat] = (graph + 8 + meg_txt + edge_txt + 3u) & ~3u;
}

// records, compacted: header (n_vertices, n_edges, flags, 0), vertices, CSR offsets (u16), targets
// (u8), then the two texts
__global__ __launch_bounds__(64)
void meg_emit_kernel(uint32_t n_pat, const MegScratch* __restrict__ scratch, const uint4* __restrict__ info,
                     const unsigned long long* __restrict__ rec_off, uint8_t* __restrict__ out) {
  const uint32_t pat = blockIdx.x * blockDim.x + threadIdx.x;
  if (pat >= n_pat) return;
  const uint4 h = info[pat];
  uint8_t* rec = out + rec_off[pat];
  uint32_t* head = (uint32_t*)rec;
  head[0] = h.x; head[1] = h.y; head[2] = h.z; head[3] = 0;
  if (h.z & PGPU_MEG_UNAVAILABLE) return;
  const MegScratch* S = scratch + pat;
  const int no = (int)h.w;
  int32_t* vt = (int32_t*)(rec + 16);
  uint16_t* first = (uint16_t*)(rec + 16 + 12 * h.x);
  uint8_t* tgt = rec + 16 + 12 * h.x + 2 * (h.x + 1);
  uint32_t e = 0;
  for (int k = 0; k < no; ++k) {
    const int v = S->order[k];
    vt[3 * k] = S->p[v]; vt[3 * k + 1] = S->t[v]; vt[3 * k + 2] = S->l[v];
    first[k] = (uint16_t)e;
    for (int q = 0; q < S->nadj[v]; ++q) tgt[e++] = S->idx_of[S->adj[v][q]];
  }
  first[no] = (uint16_t)e;
  const uint32_t graph = (16 + 12 * h.x + 2 * (h.x + 1) + h.y + 3u) & ~3u;
  for (uint8_t* z = tgt + e; z < rec + graph; ++z) *z = 0;        // the padding is part of the record: same bytes every run
  const uint32_t meg_txt = (uint32_t)(S->star[0] & 0xFFFFFFFFull), edge_txt = (uint32_t)(S->star[0] >> 32);
  uint32_t* tl = (uint32_t*)(rec + graph);
  tl[0] = meg_txt; tl[1] = edge_txt;
  uint8_t* w = rec + graph + 8;
  for (int k = 0; k < no; ++k) {
    const int v = S->order[k];
    *w++ = '('; w = put_dec(w, S->p[v]); *w++ = ','; w = put_dec(w, S->t[v]); *w++ = ','; w = put_dec(w, S->l[v]); *w++ = ')'; *w++ = '\n';
  }
  *w++ = '#'; *w++ = 'a'; *w++ = 'd'; *w++ = 'j'; *w++ = '#'; *w++ = '\n';
  for (int k = 0; k < no; ++k) {
    const int v = S->order[k];
    for (int q = 0; q < S->nadj[v]; ++q) { w = put_dec(w, k); *w++ = '-'; w = put_dec(w, (int32_t)S->idx_of[S->adj[v][q]]); *w++ = '\n'; }
  }
  for (int k = 0; k < no; ++k) {
    const int v = S->order[k];
    if (S->p[v] == SRC_START || S->p[v] == SINK_START) continue;
    for (int q = 0; q < S->nadj[v]; ++q) {
      const int x = S->adj[v][q];
      if (S->p[x] == SINK_START) continue;
      int32_t v9[9];
      edge_fields(S, v, x, v9);
      for (int f = 0; f < 9; ++f) { if (f) *w++ = ' '; w = put_dec(w, v9[f]); }
      if (v9[6] >= 50) { const char* t = " intronic"; for (int c = 0; c < 9; ++c) *w++ = (uint8_t)t[c]; }
      *w++ = '\n';
    }
  }
  while ((w - rec) & 3) *w++ = 0;
}

// ---------------------------------------------------------------------------------------------
// exclusive prefix sums (u32 counts -> u64 offsets) for the compactions of the pairing and MEG
// stages: three small kernels, wave scans by lane shuffles
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ unsigned long long wave_inclusive(unsigned long long v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(v, d);
    if (lane >= (uint32_t)d) v += o;
  }
  return v;
}

// exclusive scan of the block's SCAN_BLOCK values; returns the block total in *total
__device__ unsigned long long block_exclusive(unsigned long long v, unsigned long long* total) {
  __shared__ unsigned long long wsum[SCAN_BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned long long inc = wave_inclusive(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < SCAN_BLOCK / 64; ++k) { if ((uint32_t)k < w) base += wsum[k]; tot += wsum[k]; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK)
void scan_tile_sums_kernel(const uint32_t* __restrict__ in, size_t n, unsigned long long* __restrict__ tile_sum) {
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  unsigned long long s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) s += in[base + k];
  unsigned long long tot;
  block_exclusive(s, &tot);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK)
void scan_tile_offsets_kernel(unsigned long long* __restrict__ tile_sum, size_t n_tiles) {   // one block
  unsigned long long carry = 0;
  for (size_t at = 0; at < n_tiles; at += SCAN_BLOCK) {
    const size_t i = at + threadIdx.x;
    const unsigned long long v = i < n_tiles ? tile_sum[i] : 0;
    unsigned long long tot;
    const unsigned long long ex = block_exclusive(v, &tot);
    if (i < n_tiles) tile_sum[i] = carry + ex;
    carry += tot;
  }
}

__global__ __launch_bounds__(SCAN_BLOCK)
void scan_apply_kernel(const uint32_t* __restrict__ in, size_t n, const unsigned long long* __restrict__ tile_off,
                       unsigned long long* __restrict__ out) {
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  unsigned long long s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
  unsigned long long tot;
  unsigned long long run = tile_off[blockIdx.x] + block_exclusive(s, &tot);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

}  // namespace

size_t pgpu_scan_tmp_bytes(size_t n) { return ((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(unsigned long long); }

// out[i] = in[0] + ... + in[i-1] for i in [0, n); `tmp` holds pgpu_scan_tmp_bytes(n)
void pgpu_exclusive_scan_u32(const uint32_t* in, unsigned long long* out, size_t n, void* tmp, hipStream_t st) {
  if (n == 0) return;
  const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  unsigned long long* ts = (unsigned long long*)tmp;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, in, n, ts);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, ts, tiles);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, in, n, ts, out);
}

size_t pgpu_meg_scratch_bytes(size_t n_pat) { return (n_pat ? n_pat : 1) * sizeof(MegScratch); }

void pgpu_meg_launch_build(const pgpu_pairing* pairs, const unsigned long long* pair_first, const unsigned long long* pat_off,
                           uint32_t n_pat, const pgpu_meg_params* prm, void* scratch, void* info, uint32_t* rec_bytes,
                           hipStream_t st) {
  if (n_pat == 0) return;
  hipLaunchKernelGGL(meg_build_kernel, dim3((n_pat + 63) / 64), dim3(64), 0, st, pairs, pair_first, pat_off, n_pat, *prm,
                     (MegScratch*)scratch, (uint4*)info, rec_bytes);
}

void pgpu_meg_launch_emit(uint32_t n_pat, const void* scratch, const void* info, const unsigned long long* rec_off,
                          uint8_t* out, hipStream_t st) {
  if (n_pat == 0) return;
  hipLaunchKernelGGL(meg_emit_kernel, dim3((n_pat + 63) / 64), dim3(64), 0, st, n_pat, (const MegScratch*)scratch,
                     (const uint4*)info, rec_off, out);
}
