/* TEST INFRASTRUCTURE (tests/): a CPU stand-in for the part of the C-ABI (include/pintron_gpu.h)
 * that the est-fact host program calls, implemented with the ORACLE.  It exists so that the host
 * program's batching/fibre scheduler -- which is ordinary host code -- can be exercised in a
 * container without a GPU (tests/test_host_estfact.py).  It is linked ONLY into
 * tests/hostcheck/estfact_sched_check; the product links libpintron_gpu.so and nothing else. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pintron_gpu.h"
#include "../../pintron_amd/host/estfact.h"     /* the host's own MEG routines build the stand-in's records */
#include "../../oracle/dp_oracle.h"
#include "../../oracle/pairing_oracle.h"

uint64_t orc_dp_batch(const pgpu_dp_job* jobs, size_t n, const char* arena, const char* genomic,
                      pgpu_dp_result* results, char* strings, size_t strings_cap, size_t* strings_used);

struct pgpu_ctx { char err[64]; };
struct pgpu_index { orc_index* ix; char* gen; size_t len; };
struct pgpu_dp_plan { pgpu_dp_job* jobs; size_t n; char* arena; const pgpu_index* idx; pgpu_dp_result* res; char* strs; size_t strs_bytes; };
struct pgpu_pairing_plan { const pgpu_index* idx; char* pats; uint64_t* off; size_t n; int32_t* out; uint64_t* first; size_t cnt;
                           unsigned char* meg; uint64_t* meg_first; size_t meg_bytes; };

int pgpu_init(int device, pgpu_ctx** ctx) { (void)device; *ctx = (pgpu_ctx*)calloc(1, sizeof(pgpu_ctx)); return PGPU_OK; }
int pgpu_destroy(pgpu_ctx* ctx) { free(ctx); return PGPU_OK; }
const char* pgpu_last_error(const pgpu_ctx* ctx) { (void)ctx; return "fake"; }
int pgpu_index_build(pgpu_ctx* ctx, const char* g, size_t len, pgpu_index** idx) {
  (void)ctx;
  pgpu_index* x = (pgpu_index*)calloc(1, sizeof(*x));
  x->gen = (char*)calloc(len + 8, 1); memcpy(x->gen, g, len); x->len = len;
  x->ix = orc_index_create(g, len);
  *idx = x; return PGPU_OK;
}
int pgpu_index_save(pgpu_ctx* ctx, const pgpu_index* idx, const char* g, const char* path) { (void)ctx; (void)idx; (void)g; (void)path; return PGPU_ENOSYS; }
int pgpu_index_load(pgpu_ctx* ctx, const char* path, const char* g, size_t len, pgpu_index** idx) { (void)ctx; (void)path; (void)g; (void)len; (void)idx; return PGPU_EINVAL; }
int pgpu_index_destroy(pgpu_ctx* ctx, pgpu_index* idx) { (void)ctx; orc_index_destroy(idx->ix); free(idx->gen); free(idx); return PGPU_OK; }

int pgpu_pairing_plan_create(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns, const uint64_t* off, size_t n, pgpu_pairing_plan** out) {
  (void)ctx;
  pgpu_pairing_plan* p = (pgpu_pairing_plan*)calloc(1, sizeof(*p));
  p->idx = idx; p->n = n;
  p->pats = (char*)malloc(off[n] + 1); memcpy(p->pats, patterns, off[n]);
  p->off = (uint64_t*)malloc((n + 1) * sizeof(uint64_t)); memcpy(p->off, off, (n + 1) * sizeof(uint64_t));
  *out = p; return PGPU_OK;
}
int pgpu_pairing_plan_create_resident(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns, const uint64_t* off, size_t n, pgpu_pairing_plan** out) {
  return pgpu_pairing_plan_create(ctx, idx, patterns, off, n, out);     /* the stand-in has no scratch to share */
}
int pgpu_pairing_plan_run(pgpu_ctx* ctx, pgpu_pairing_plan* p, const pgpu_pairing_params* prm) {
  (void)ctx;
  if (getenv("PINTRON_FAKE_CACHE") && p->out && p->first) return PGPU_OK;   /* profiling aid: see below */
  free(p->out); free(p->first);
  size_t cap = 1 << 16;
  p->out = (int32_t*)malloc(3 * cap * sizeof(int32_t));
  p->first = (uint64_t*)malloc((p->n + 1) * sizeof(uint64_t));
  size_t tot = 0;
  for (size_t i = 0; i < p->n; ++i) {
    p->first[i] = tot;
    const size_t m = (size_t)(p->off[i + 1] - p->off[i]);
    char* pat = (char*)malloc(m + 1); memcpy(pat, p->pats + p->off[i], m); pat[m] = '\0';
    long c = orc_pairings(p->idx->ix, pat, m, prm->min_factor_len, prm->min_string_depth_rate, p->out + 3 * tot, (long)(cap - tot), NULL);
    if ((size_t)c > cap - tot) {
      cap = (tot + (size_t)c) * 2; p->out = (int32_t*)realloc(p->out, 3 * cap * sizeof(int32_t));
      c = orc_pairings(p->idx->ix, pat, m, prm->min_factor_len, prm->min_string_depth_rate, p->out + 3 * tot, (long)(cap - tot), NULL);
    }
    free(pat);
    tot += (size_t)c;
  }
  p->first[p->n] = tot; p->cnt = tot;
  return PGPU_OK;
}
uint64_t pgpu_pairing_plan_count(const pgpu_pairing_plan* p) { return p->cnt; }
uint64_t pgpu_pairing_plan_positions(const pgpu_pairing_plan* p) { return p->off[p->n]; }
int pgpu_pairing_plan_fetch(pgpu_ctx* ctx, pgpu_pairing_plan* p, pgpu_pairing* out, size_t cap, uint64_t* first) {
  (void)ctx;
  if (cap < p->cnt) return PGPU_ENOSPC;
  memcpy(out, p->out, p->cnt * sizeof(pgpu_pairing));
  memcpy(first, p->first, (p->n + 1) * sizeof(uint64_t));
  return PGPU_OK;
}
int pgpu_pairing_plan_destroy(pgpu_ctx* ctx, pgpu_pairing_plan* p) { (void)ctx; free(p->pats); free(p->off); free(p->out); free(p->first); free(p->meg); free(p->meg_first); free(p); return PGPU_OK; }

/* MEG stage of the stand-in: the host's own (reference-checked) MEG code, serialised in the record
 * layout of include/pintron_gpu.h.  PINTRON_FAKE_NO_MEG=1 answers PGPU_ENOSYS instead (a library
 * without the stage); PINTRON_FAKE_MEG_LIMIT=<n> flags graphs with more vertices as unavailable. */
int pgpu_pairing_plan_run_meg(pgpu_ctx* ctx, pgpu_pairing_plan* p, const pgpu_meg_params* mp) {
  (void)ctx;
  if (getenv("PINTRON_FAKE_NO_MEG")) return PGPU_ENOSYS;
  if (getenv("PINTRON_FAKE_CACHE") && p->meg) return PGPU_OK;
  const char* lim = getenv("PINTRON_FAKE_MEG_LIMIT");
  const size_t max_v = lim ? (size_t)atol(lim) : PGPU_MEG_MAX_VERTICES;
  ef_config cfg; memset(&cfg, 0, sizeof cfg);
  cfg.min_factor_len = mp->min_factor_len; cfg.min_intron_length = mp->min_intron_length; cfg.max_intron_length = mp->max_intron_length;
  cfg.max_pairings_in_MEG = mp->max_pairings_in_MEG; cfg.max_prefix_discarded_rate = mp->max_prefix_discarded_rate;
  cfg.max_suffix_discarded_rate = mp->max_suffix_discarded_rate; cfg.max_freq_shortest_pairing = mp->max_freq_shortest_pairing;
  cfg.trans_red = mp->trans_red != 0; cfg.short_edge_comp = mp->short_edge_comp != 0;
  free(p->meg); free(p->meg_first);
  size_t cap = 1 << 16, len = 0;
  p->meg = (unsigned char*)malloc(cap);
  p->meg_first = (uint64_t*)malloc((p->n + 1) * sizeof(uint64_t));
  for (size_t i = 0; i < p->n; ++i) {
    p->meg_first[i] = len;
    const size_t m = (size_t)(p->off[i + 1] - p->off[i]);
    ef_meg* V = ef_meg_from_pairings((const ef_triple*)(p->out + 3 * p->first[i]), (size_t)(p->first[i + 1] - p->first[i]), m);
    ef_build_edge_set(V, &cfg);
    ef_simplify_meg(V, &cfg);
    if (cfg.trans_red) ef_transitive_reduction(V);
    bool complex = ef_is_too_complex_for_compaction(V);
    if (!complex && cfg.short_edge_comp) ef_compact_short_edges(V, &cfg);
    complex = complex || ef_is_too_complex(V, &cfg);
    size_t tp, te;
    ef_meg_stats(V, &tp, &te);
    const bool unavailable = tp > max_v || tp > 255 || te > 60000;
    const size_t graph = (16 + 12 * tp + 2 * (tp + 1) + te + 3) & ~(size_t)3;
    ef_sink t1 = { NULL, NULL, 0, 0 }, t2 = { NULL, NULL, 0, 0 };
    if (!unavailable) { ef_meg_write(&t1, V); ef_intronic_edges_write(&t2, V); }
    const size_t bytes = unavailable ? 16 : ((graph + 8 + t1.len + t2.len + 3) & ~(size_t)3);
    if (len + bytes > cap) { cap = (len + bytes) * 2; p->meg = (unsigned char*)realloc(p->meg, cap); }
    unsigned char* rec = p->meg + len;
    memset(rec, 0, bytes);
    uint32_t* head = (uint32_t*)rec;
    if (unavailable) head[2] = PGPU_MEG_UNAVAILABLE;
    else {
      head[0] = (uint32_t)tp; head[1] = (uint32_t)te; head[2] = complex ? PGPU_MEG_TOO_COMPLEX : 0;
      int32_t* vt = (int32_t*)(rec + 16);
      uint16_t* first = (uint16_t*)(rec + 16 + 12 * tp);
      unsigned char* tgt = rec + 16 + 12 * tp + 2 * (tp + 1);
      int k = 0;
      EF_MEG_FOR_POS(V, pos, 0, V->n) {
        ef_iter it = efl_begin(V->v[pos]);
        while (efi_has_next(&it)) { ef_pairing* q = (ef_pairing*)efi_next(&it); q->id = k; vt[3 * k] = q->p; vt[3 * k + 1] = q->t; vt[3 * k + 2] = q->l; ++k; }
      }
      uint32_t e = 0; k = 0;
      EF_MEG_FOR_POS(V, pos, 0, V->n) {
        ef_iter it = efl_begin(V->v[pos]);
        while (efi_has_next(&it)) {
          ef_pairing* q = (ef_pairing*)efi_next(&it);
          first[k++] = (uint16_t)e;
          ef_iter a = efl_begin(q->adjs);
          while (efi_has_next(&a)) tgt[e++] = (unsigned char)((ef_pairing*)efi_next(&a))->id;
        }
      }
      first[k] = (uint16_t)e;
      uint32_t* tl = (uint32_t*)(rec + graph);
      tl[0] = (uint32_t)t1.len; tl[1] = (uint32_t)t2.len;
      memcpy(rec + graph + 8, t1.mem, t1.len); memcpy(rec + graph + 8 + t1.len, t2.mem, t2.len);
    }
    free(t1.mem); free(t2.mem);
    ef_meg_free(V);
    len += bytes;
  }
  p->meg_first[p->n] = len; p->meg_bytes = len;
  return PGPU_OK;
}
uint64_t pgpu_pairing_plan_meg_bytes(const pgpu_pairing_plan* p) { return p->meg_bytes; }
double pgpu_pairing_plan_meg_ms(const pgpu_pairing_plan* p) { (void)p; return 0.0; }
int pgpu_pairing_plan_fetch_meg(pgpu_ctx* ctx, pgpu_pairing_plan* p, void* out, size_t cap, uint64_t* first) {
  (void)ctx;
  if (cap < p->meg_bytes) return PGPU_ENOSPC;
  memcpy(out, p->meg, p->meg_bytes);
  memcpy(first, p->meg_first, (p->n + 1) * sizeof(uint64_t));
  return PGPU_OK;
}
/* gather of the stand-in: the ranks are processes on one host, the "wire" is a directory of files
 * named after the communicator id (which rank 0 made and handed over exactly as with RCCL) */
#include <stdio.h>
#include <time.h>
#include <unistd.h>
struct pgpu_comm { char tag[64]; int rank, world; unsigned seq; };
int pgpu_comm_unique_id(pgpu_ctx* ctx, pgpu_comm_id* id) {
  (void)ctx;
  memset(id, 0, sizeof *id);
  snprintf(id->bytes, sizeof id->bytes, "%ld-%ld", (long)getpid(), (long)time(NULL));
  return PGPU_OK;
}
int pgpu_comm_init(pgpu_ctx* ctx, int rank, int world, const pgpu_comm_id* id, pgpu_comm** out) {
  (void)ctx;
  pgpu_comm* c = (pgpu_comm*)calloc(1, sizeof *c);
  snprintf(c->tag, sizeof c->tag, "%.60s", id->bytes); c->rank = rank; c->world = world;
  *out = c; return PGPU_OK;
}
int pgpu_comm_destroy(pgpu_ctx* ctx, pgpu_comm* c) { (void)ctx; free(c); return PGPU_OK; }
int pgpu_gather(pgpu_ctx* ctx, pgpu_comm* c, const void* send, uint64_t send_bytes, void* recv, uint64_t recv_cap, uint64_t* counts) {
  (void)ctx;
  const char* dir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
  char path[512], tmp[520];
  const unsigned seq = c->seq++;
  if (seq >= 2) {             /* every rank is past round seq-2 by now: its marker can go */
    snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-seen-%d", dir, c->tag, seq - 2, c->rank);
    unlink(path);
  }
  snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-%d", dir, c->tag, seq, c->rank);
  snprintf(tmp, sizeof tmp, "%s.tmp", path);
  FILE* f = fopen(tmp, "wb");
  if (!f) return PGPU_EDEVICE;
  if (send_bytes) fwrite(send, 1, send_bytes, f);
  fclose(f);
  rename(tmp, path);
  uint64_t at = 0;
  int rc = PGPU_OK;
  for (int r = 0; r < c->world; ++r) {                  /* every rank reads every size; rank 0 the data too */
    snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-%d", dir, c->tag, seq, r);
    FILE* g = NULL;
    for (int tries = 0; tries < 30000 && !g; ++tries) { g = fopen(path, "rb"); if (!g) { struct timespec ts = { 0, 2000000 }; nanosleep(&ts, NULL); } }
    if (!g) return PGPU_EDEVICE;
    fseek(g, 0, SEEK_END); counts[r] = (uint64_t)ftell(g); fseek(g, 0, SEEK_SET);
    if (c->rank == 0 && rc == PGPU_OK) {
      if (at + counts[r] > recv_cap) rc = PGPU_ENOSPC;
      else if (counts[r] && fread((char*)recv + at, 1, counts[r], g) != counts[r]) rc = PGPU_EDEVICE;
      at += counts[r];
    }
    fclose(g);
  }
  /* a second rendezvous so that nobody removes a file somebody has not read yet: each rank drops a
   * "seen" marker, rank r removes its own payload once all markers are there */
  snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-seen-%d", dir, c->tag, seq, c->rank);
  f = fopen(path, "wb"); if (f) fclose(f);
  for (int r = 0; r < c->world; ++r) {
    snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-seen-%d", dir, c->tag, seq, r);
    for (int tries = 0; tries < 30000 && access(path, F_OK) != 0; ++tries) { struct timespec ts = { 0, 2000000 }; nanosleep(&ts, NULL); }
  }
  snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-%d", dir, c->tag, seq, c->rank);
  unlink(path);
  return rc;
}

/* fixed-size all-gather over the same "wire": a gather to a scratch buffer on every rank */
int pgpu_allgather(pgpu_ctx* ctx, pgpu_comm* c, const void* send, uint64_t bytes, void* recv) {
  (void)ctx;
  const char* dir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
  char path[512], tmp[520];
  const unsigned seq = c->seq++;
  if (seq >= 2) { snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-seen-%d", dir, c->tag, seq - 2, c->rank); unlink(path); }
  snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-%d", dir, c->tag, seq, c->rank);
  snprintf(tmp, sizeof tmp, "%s.tmp", path);
  FILE* f = fopen(tmp, "wb");
  if (!f) return PGPU_EDEVICE;
  if (bytes) fwrite(send, 1, bytes, f);
  fclose(f);
  rename(tmp, path);
  for (int r = 0; r < c->world; ++r) {
    snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-%d", dir, c->tag, seq, r);
    FILE* g = NULL;
    for (int tries = 0; tries < 30000 && !g; ++tries) { g = fopen(path, "rb"); if (!g) { struct timespec ts = { 0, 2000000 }; nanosleep(&ts, NULL); } }
    if (!g) return PGPU_EDEVICE;
    const int bad = bytes && fread((char*)recv + (size_t)r * bytes, 1, bytes, g) != bytes;
    fclose(g);
    if (bad) return PGPU_EDEVICE;
  }
  snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-seen-%d", dir, c->tag, seq, c->rank);
  f = fopen(path, "wb"); if (f) fclose(f);
  for (int r = 0; r < c->world; ++r) {
    snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-seen-%d", dir, c->tag, seq, r);
    for (int tries = 0; tries < 30000 && access(path, F_OK) != 0; ++tries) { struct timespec ts = { 0, 2000000 }; nanosleep(&ts, NULL); }
  }
  snprintf(path, sizeof path, "%s/pgpu-fake-%s-%u-%d", dir, c->tag, seq, c->rank);
  unlink(path);
  return PGPU_OK;
}

void pgpu_range_push(const char* name) { (void)name; }
void pgpu_range_pop(void) { }
const char* pgpu_build_info(void) { return "CPU stand-in of the C-ABI (tests/hostcheck/fake_pgpu.c)"; }
int pgpu_host_alloc(pgpu_ctx* ctx, size_t bytes, void** out) { (void)ctx; *out = malloc(bytes ? bytes : 16); return *out ? PGPU_OK : PGPU_ENOMEM; }
int pgpu_host_free(pgpu_ctx* ctx, void* q) { (void)ctx; free(q); return PGPU_OK; }
int pgpu_pairings(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns, const uint64_t* off, size_t n, const pgpu_pairing_params* prm,
                  pgpu_pairing* out, size_t cap, uint64_t* first, size_t* n_out) {
  pgpu_pairing_plan* p; pgpu_pairing_plan_create(ctx, idx, patterns, off, n, &p);
  pgpu_pairing_plan_run(ctx, p, prm);
  if (n_out) *n_out = p->cnt;
  const int rc = pgpu_pairing_plan_fetch(ctx, p, out, cap, first);
  pgpu_pairing_plan_destroy(ctx, p);
  return rc;
}

int pgpu_dp_plan_create(pgpu_ctx* ctx, const pgpu_index* idx, const pgpu_dp_job* jobs, size_t n, const char* arena, size_t alen, pgpu_dp_plan** out) {
  (void)ctx;
  pgpu_dp_plan* p = (pgpu_dp_plan*)calloc(1, sizeof(*p));
  p->jobs = (pgpu_dp_job*)malloc((n + 1) * sizeof(pgpu_dp_job)); memcpy(p->jobs, jobs, n * sizeof(pgpu_dp_job));
  p->n = n; p->arena = (char*)calloc(alen + 8, 1); memcpy(p->arena, arena, alen); p->idx = idx;
  for (size_t i = 0; i < n; ++i) if (jobs[i].kind <= PGPU_DP_GAP) p->strs_bytes += 2 * ((size_t)jobs[i].a_len + jobs[i].b_len + 1);
  if (getenv("PINTRON_FAKE_STATS")) {    /* one line per batch and kind: jobs, longest sweep (rows + columns) and its shape */
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    static FILE* f; if (!f) f = fopen(getenv("PINTRON_FAKE_STATS"), "w");
    for (uint32_t k = 0; k < 7; ++k) {
      size_t cnt = 0, best = 0, br = 0, bc = 0, big = 0;
      for (size_t i = 0; i < n; ++i) if (jobs[i].kind == k) {
        size_t r = jobs[i].a_len, c = jobs[i].b_len;
        if (k == PGPU_DP_BORDERS) c = r + jobs[i].p2 < c ? r + jobs[i].p2 : c;
        if (k == PGPU_DP_ED || k == PGPU_DP_KBAND) { if (r > c) { size_t t = r; r = c; c = t; } }
        ++cnt; if (r > 64) ++big;
        if (r + c >= best) { best = r + c; br = r; bc = c; }
      }
      if (cnt) fprintf(f, "%u %zu %zu %zu %zu %zu\n", k, cnt, big, best, br, bc);
    }
    pthread_mutex_unlock(&mu);
  }
  *out = p; return PGPU_OK;
}
/* PINTRON_FAKE_CACHE=1 (profiling aid for tests/hostcheck/sched_profile): answers are remembered by
 * request content, so a second pass over the same batch costs almost no DP time and what remains
 * is the host logic plus the scheduler. */
typedef struct cache_ent { uint64_t key; pgpu_dp_result res; char* s0; char* s1; struct cache_ent* next; } cache_ent;
#define CACHE_BUCKETS (1u << 20)
static cache_ent** cache_tab;
static pthread_mutex_t cache_mu = PTHREAD_MUTEX_INITIALIZER;
static uint64_t fnv(uint64_t h, const void* p, size_t n) {
  const unsigned char* c = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
  return h;
}
static void cached_job(const pgpu_dp_plan* p, size_t i, char* strs, size_t* spos) {
  const pgpu_dp_job* j = &p->jobs[i];
  const char* gen = p->idx ? p->idx->gen : NULL;
  const char* a = ((j->flags & PGPU_JOB_A_GENOMIC) ? gen : p->arena) + j->a_off;
  const char* b = ((j->flags & PGPU_JOB_B_GENOMIC) ? gen : p->arena) + j->b_off;
  uint64_t h = 1469598103934665603ull;
  h = fnv(h, &j->kind, 4); h = fnv(h, &j->a_len, 4); h = fnv(h, &j->b_len, 4);
  h = fnv(h, &j->p0, 16); h = fnv(h, &j->flags, 4);
  /* an operand inside the genomic sequence is identified by its offset */
  if (j->flags & PGPU_JOB_A_GENOMIC) h = fnv(h, &j->a_off, 8); else h = fnv(h, a, j->a_len);
  if (j->flags & PGPU_JOB_B_GENOMIC) h = fnv(h, &j->b_off, 8); else h = fnv(h, b, j->b_len + (j->tail > 2 ? 2 : j->tail));
  pthread_mutex_lock(&cache_mu);
  if (!cache_tab) cache_tab = (cache_ent**)calloc(CACHE_BUCKETS, sizeof(cache_ent*));
  cache_ent* e = cache_tab[h & (CACHE_BUCKETS - 1)];
  while (e && e->key != h) e = e->next;
  pthread_mutex_unlock(&cache_mu);
  if (!e) {
    e = (cache_ent*)calloc(1, sizeof(cache_ent));
    e->key = h;
    char* tmp = (char*)malloc(2 * ((size_t)j->a_len + j->b_len + 1) + 16);
    size_t used;
    orc_dp_batch(j, 1, p->arena, gen, &e->res, tmp, 2 * ((size_t)j->a_len + j->b_len + 1) + 16, &used);
    if (j->kind <= PGPU_DP_GAP && e->res.status == PGPU_OK) { e->s0 = strdup(tmp + e->res.str[0]); e->s1 = strdup(tmp + e->res.str[1]); }
    free(tmp);
    pthread_mutex_lock(&cache_mu);
    e->next = cache_tab[h & (CACHE_BUCKETS - 1)]; cache_tab[h & (CACHE_BUCKETS - 1)] = e;
    pthread_mutex_unlock(&cache_mu);
  }
  p->res[i] = e->res;
  if (e->s0) {
    const size_t l0 = strlen(e->s0) + 1, l1 = strlen(e->s1) + 1;
    memcpy(strs + *spos, e->s0, l0); p->res[i].str[0] = *spos; *spos += l0;
    memcpy(strs + *spos, e->s1, l1); p->res[i].str[1] = *spos; *spos += l1;
  }
}

int pgpu_dp_plan_create_parts(pgpu_ctx* ctx, const pgpu_index* idx, const pgpu_dp_part* parts, size_t n_parts, pgpu_dp_plan** out) {
  size_t nj = 0, na = 0;
  for (size_t q = 0; q < n_parts; ++q) { nj += parts[q].n_jobs; na += parts[q].arena_len; }
  pgpu_dp_job* jobs = (pgpu_dp_job*)malloc((nj + 1) * sizeof(pgpu_dp_job));
  char* arena = (char*)malloc(na + 8);
  size_t jp = 0, ap = 0;
  for (size_t q = 0; q < n_parts; ++q) {
    memcpy(jobs + jp, parts[q].jobs, parts[q].n_jobs * sizeof(pgpu_dp_job));
    memcpy(arena + ap, parts[q].arena, parts[q].arena_len);
    for (size_t i = 0; i < parts[q].n_jobs; ++i) {
      if (!(jobs[jp + i].flags & PGPU_JOB_A_GENOMIC)) jobs[jp + i].a_off += ap;
      if (!(jobs[jp + i].flags & PGPU_JOB_B_GENOMIC)) jobs[jp + i].b_off += ap;
    }
    jp += parts[q].n_jobs; ap += parts[q].arena_len;
  }
  const int rc = pgpu_dp_plan_create(ctx, idx, jobs, nj, arena, na, out);
  free(jobs); free(arena);
  return rc;
}

int pgpu_dp_plan_launch(pgpu_ctx* ctx, pgpu_dp_plan* p) {
  (void)ctx;
  p->res = (pgpu_dp_result*)malloc((p->n + 1) * sizeof(pgpu_dp_result));
  p->strs = (char*)malloc(p->strs_bytes + 16);
  size_t used;
  if (getenv("PINTRON_FAKE_CACHE")) {
    size_t spos = 0;
    for (size_t i = 0; i < p->n; ++i) cached_job(p, i, p->strs, &spos);
    return PGPU_OK;
  }
  orc_dp_batch(p->jobs, p->n, p->arena, p->idx ? p->idx->gen : NULL, p->res, p->strs, p->strs_bytes + 16, &used);
  return PGPU_OK;
}
int pgpu_dp_plan_sync(pgpu_ctx* ctx, pgpu_dp_plan* p) { (void)ctx; (void)p; return PGPU_OK; }
size_t pgpu_dp_plan_string_bytes(const pgpu_dp_plan* p) { return p->strs_bytes; }
int pgpu_dp_plan_fetch(pgpu_ctx* ctx, pgpu_dp_plan* p, pgpu_dp_result* res, char* strings, size_t cap) {
  (void)ctx;
  if (cap < p->strs_bytes) return PGPU_ENOSPC;
  memcpy(res, p->res, p->n * sizeof(pgpu_dp_result));
  if (p->strs_bytes) memcpy(strings, p->strs, p->strs_bytes);
  return PGPU_OK;
}
int pgpu_dp_plan_destroy(pgpu_ctx* ctx, pgpu_dp_plan* p) { (void)ctx; free(p->jobs); free(p->arena); free(p->res); free(p->strs); free(p); return PGPU_OK; }
int pgpu_dp_batch(pgpu_ctx* ctx, const pgpu_index* idx, const pgpu_dp_job* jobs, size_t n, const char* arena, size_t alen,
                  pgpu_dp_result* res, char* strings, size_t cap, size_t* used) {
  pgpu_dp_plan* p; pgpu_dp_plan_create(ctx, idx, jobs, n, arena, alen, &p);
  pgpu_dp_plan_launch(ctx, p);
  if (used) *used = p->strs_bytes;
  const int rc = pgpu_dp_plan_fetch(ctx, p, res, strings, cap);
  pgpu_dp_plan_destroy(ctx, p);
  return rc;
}

int pgpu_set_timing(pgpu_ctx* ctx, int enabled) { (void)ctx; (void)enabled; return PGPU_OK; }
int pgpu_device_numa_node(pgpu_ctx* ctx) { (void)ctx; return -1; }
int pgpu_dp_plan_n_groups(const pgpu_dp_plan* p) { (void)p; return 0; }
int pgpu_dp_plan_group_info(const pgpu_dp_plan* p, int i, pgpu_group_info* out) { (void)p; (void)i; (void)out; return PGPU_EINVAL; }
double pgpu_pairing_plan_kernel_ms(const pgpu_pairing_plan* p, int k) { (void)p; (void)k; return 0.0; }
