/* GPU backend of the est-fact host program: everything computational goes through the C-ABI of
 * libpintron_gpu.so (include/pintron_gpu.h).  There is no CPU implementation behind it: when the
 * library or a gfx950 device is missing, est-fact stops with an error.
 *
 * Two modes share the request encoding:
 *   - direct:  one C-ABI call per request (ef_gpu_open / ef_gpu_close); simple, used by the tests;
 *   - batched: requests of many EST fibres are collected and submitted together (ef_sched.c).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "estfact.h"
#include "ef_gpu.h"

/* ---- PINTRON_DP_TRACE=<file>: every answered request with its answer ---------------------------------
 * A debugging aid for parity hunts: tools/replay_dp_trace.py runs the recorded requests through the CPU
 * oracle and names the answers that differ, which tells "a kernel gave a wrong answer" from "the host
 * logic / the scheduler went wrong".  Record (little endian): u32 magic 'DPT1', u32 unit, i32 kind,
 * u32 la, lb, p0, p1, p2, tail, i32 v[6], u32 n0, n1, then a[la], b[lb + min(tail, 2)], s0[n0], s1[n1]. */
static FILE* trace_file;
static pthread_mutex_t trace_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_once_t trace_once = PTHREAD_ONCE_INIT;
static void trace_open(void) {
  const char* path = getenv("PINTRON_DP_TRACE");
  if (path && path[0]) trace_file = fopen(path, "wb");
}
int ef_dp_trace_enabled(void) { pthread_once(&trace_once, trace_open); return trace_file != NULL; }
void ef_dp_trace(const ef_dp_req* q, const ef_dp_res* r, uint32_t unit) {
  if (!ef_dp_trace_enabled()) return;
  const uint32_t tail = q->tail > 2 ? 2 : q->tail;
  const uint32_t n0 = r->s0 ? (uint32_t)strlen(r->s0) : 0, n1 = r->s1 ? (uint32_t)strlen(r->s1) : 0;
  const uint32_t head[9] = { 0x31545044u, unit, (uint32_t)q->kind, (uint32_t)q->la, (uint32_t)q->lb, q->p0, q->p1, q->p2, q->tail };
  const uint32_t lens[2] = { n0, n1 };
  pthread_mutex_lock(&trace_mu);
  fwrite(head, 4, 9, trace_file); fwrite(r->v, 4, 6, trace_file); fwrite(lens, 4, 2, trace_file);
  fwrite(q->a, 1, q->la, trace_file); fwrite(q->b, 1, q->lb + tail, trace_file);
  if (n0) fwrite(r->s0, 1, n0, trace_file);
  if (n1) fwrite(r->s1, 1, n1, trace_file);
  pthread_mutex_unlock(&trace_mu);
}
void ef_dp_trace_flush(void) { if (trace_file) { pthread_mutex_lock(&trace_mu); fflush(trace_file); pthread_mutex_unlock(&trace_mu); } }

int ef_gpu_device_from_env(void) {
  const char* d = getenv("PINTRON_GPU_DEVICE");
  return d ? atoi(d) : 0;
}

/* ---- request encoding --------------------------------------------------------------------------- */
void ef_jobbuf_init(ef_jobbuf* jb) { memset(jb, 0, sizeof(*jb)); }

void ef_jobbuf_reset(ef_jobbuf* jb) { jb->n = 0; jb->arena_len = 0; }

void ef_jobbuf_free(ef_jobbuf* jb) { free(jb->jobs); free(jb->arena); memset(jb, 0, sizeof(*jb)); }

static uint64_t put(ef_jobbuf* jb, const char* s, size_t n) {
  if (jb->arena_len + n + 8 > jb->arena_cap) {
    jb->arena_cap = (jb->arena_len + n + 8) * 2 + 4096;
    jb->arena = (char*)realloc(jb->arena, jb->arena_cap);
  }
  const uint64_t off = jb->arena_len;
  memcpy(jb->arena + off, s, n);
  jb->arena_len += n;
  return off;
}

/* appends one request; operands that lie inside the genomic sequence are passed by offset */
size_t ef_jobbuf_add(ef_jobbuf* jb, const ef_dp_req* q, const char* gen, size_t gen_len) {
  if (jb->n == jb->cap) { jb->cap = jb->cap ? jb->cap * 2 : 256; jb->jobs = (pgpu_dp_job*)realloc(jb->jobs, jb->cap * sizeof(pgpu_dp_job)); }
  pgpu_dp_job* j = &jb->jobs[jb->n];
  memset(j, 0, sizeof(*j));
  j->kind = (uint32_t)q->kind;
  j->a_len = (uint32_t)q->la; j->b_len = (uint32_t)q->lb;
  j->p0 = q->p0; j->p1 = q->p1; j->p2 = q->p2; j->tail = q->tail;
  if (q->a >= gen && q->a + q->la <= gen + gen_len) { j->flags |= PGPU_JOB_A_GENOMIC; j->a_off = (uint64_t)(q->a - gen); }
  else j->a_off = put(jb, q->a, q->la);
  if (q->b >= gen && q->b + q->lb + q->tail <= gen + gen_len) { j->flags |= PGPU_JOB_B_GENOMIC; j->b_off = (uint64_t)(q->b - gen); }
  else j->b_off = put(jb, q->b, q->lb + (q->tail > 2 ? 2 : q->tail));
  return jb->n++;
}

/* pgpu_dp_result -> ef_dp_res (alignment rows are copied out of the batch's string buffer into one
 * padded block, see ef_dp_res) */
int ef_decode_result(int kind, const pgpu_dp_result* r, const char* strings, ef_dp_res* out) {
  memset(out, 0, sizeof(*out));
  if (r->status != PGPU_OK) return r->status;
  for (int k = 0; k < 6; ++k) out->v[k] = r->v[k];
  if (kind == EF_DP_ALIGN || kind == EF_DP_GAP) {
    const char* row0 = strings + r->str[0]; const char* row1 = strings + r->str[1];
    const size_t n0 = strlen(row0), n1 = strlen(row1);
    char* blk = (char*)malloc(n0 + n1 + 2 * EF_ROW_PAD);
    memcpy(blk, row0, n0); memset(blk + n0, 0, EF_ROW_PAD);
    memcpy(blk + n0 + EF_ROW_PAD, row1, n1); memset(blk + n0 + EF_ROW_PAD + n1, 0, EF_ROW_PAD);
    out->s0 = blk; out->s1 = blk + n0 + EF_ROW_PAD;
  }
  return 0;
}

/* ---- direct mode --------------------------------------------------------------------------------- */
typedef struct {
  pgpu_ctx* ctx;
  pgpu_index* idx;
  const char* gen; size_t gen_len;
  ef_jobbuf jb;
  char* strings; size_t strings_cap;
} direct_be;

static int direct_pairings(void* self, const char* pattern, size_t m, unsigned L, double rate, ef_triple** out, size_t* n) {
  direct_be* d = (direct_be*)self;
  const uint64_t off[2] = { 0, m };
  pgpu_pairing_params prm = { L, 0, rate };
  size_t cap = 4096, cnt = 0;
  uint64_t first[2];
  pgpu_pairing* buf = (pgpu_pairing*)malloc(cap * sizeof(pgpu_pairing));
  int rc = pgpu_pairings(d->ctx, d->idx, pattern, off, 1, &prm, buf, cap, first, &cnt);
  if (rc == PGPU_ENOSPC) {
    cap = cnt; buf = (pgpu_pairing*)realloc(buf, cap * sizeof(pgpu_pairing));
    rc = pgpu_pairings(d->ctx, d->idx, pattern, off, 1, &prm, buf, cap, first, &cnt);
  }
  if (rc != PGPU_OK) { fprintf(stderr, "* FATAL pgpu_pairings: %s\n", pgpu_last_error(d->ctx)); free(buf); return -1; }
  *out = (ef_triple*)buf; *n = cnt;          /* same layout: three int32 */
  return 0;
}

static int direct_dp(void* self, const ef_dp_req* q, ef_dp_res* res) {
  direct_be* d = (direct_be*)self;
  ef_jobbuf_reset(&d->jb);
  ef_jobbuf_add(&d->jb, q, d->gen, d->gen_len);
  const size_t need = 2 * (q->la + q->lb + 1) + 16;
  if (need > d->strings_cap) { d->strings_cap = need * 2; d->strings = (char*)realloc(d->strings, d->strings_cap); }
  pgpu_dp_result r;
  size_t used = 0;
  const int rc = pgpu_dp_batch(d->ctx, d->idx, d->jb.jobs, 1, d->jb.arena, d->jb.arena_len, &r, d->strings, d->strings_cap, &used);
  if (rc != PGPU_OK) { fprintf(stderr, "* FATAL pgpu_dp_batch: %s\n", pgpu_last_error(d->ctx)); return -1; }
  if (r.status != PGPU_OK) { fprintf(stderr, "* FATAL DP job of kind %d and size %zu x %zu exceeds the device limits\n", q->kind, q->la, q->lb); return -1; }
  const int drc = ef_decode_result(q->kind, &r, d->strings, res);
  if (drc == 0) ef_dp_trace(q, res, 0);
  return drc;
}

static int direct_dp_many(void* self, const ef_dp_req* reqs, ef_dp_res* res, size_t n) {
  direct_be* d = (direct_be*)self;
  if (n == 0) return 0;
  ef_jobbuf_reset(&d->jb);
  size_t need = 16;
  for (size_t k = 0; k < n; ++k) { ef_jobbuf_add(&d->jb, &reqs[k], d->gen, d->gen_len); need += 2 * (reqs[k].la + reqs[k].lb + 1); }
  if (need > d->strings_cap) { d->strings_cap = need * 2; d->strings = (char*)realloc(d->strings, d->strings_cap); }
  pgpu_dp_result* r = (pgpu_dp_result*)malloc(n * sizeof(pgpu_dp_result));
  size_t used = 0;
  int rc = pgpu_dp_batch(d->ctx, d->idx, d->jb.jobs, n, d->jb.arena, d->jb.arena_len, r, d->strings, d->strings_cap, &used);
  if (rc != PGPU_OK) fprintf(stderr, "* FATAL pgpu_dp_batch: %s\n", pgpu_last_error(d->ctx));
  for (size_t k = 0; k < n && rc == PGPU_OK; ++k) {
    if (r[k].status != PGPU_OK) { fprintf(stderr, "* FATAL DP job of kind %d and size %zu x %zu exceeds the device limits\n", reqs[k].kind, reqs[k].la, reqs[k].lb); rc = -1; }
    else { rc = ef_decode_result(reqs[k].kind, &r[k], d->strings, &res[k]); if (rc == PGPU_OK) ef_dp_trace(&reqs[k], &res[k], 0); }
  }
  free(r);
  return rc == PGPU_OK ? 0 : -1;
}

ef_backend* ef_gpu_open(const ef_seq* gen) {
  direct_be* d = (direct_be*)calloc(1, sizeof(direct_be));
  if (pgpu_init(ef_gpu_device_from_env(), &d->ctx) != PGPU_OK) { free(d); return NULL; }
  d->gen = gen->seq; d->gen_len = strlen(gen->seq);
  if (pgpu_index_build(d->ctx, d->gen, d->gen_len, &d->idx) != PGPU_OK) {
    fprintf(stderr, "* FATAL pgpu_index_build: %s\n", pgpu_last_error(d->ctx));
    pgpu_destroy(d->ctx); free(d);
    return NULL;
  }
  ef_jobbuf_init(&d->jb);
  ef_backend* be = (ef_backend*)calloc(1, sizeof(ef_backend));
  be->self = d; be->pairings = direct_pairings; be->dp = direct_dp; be->dp_many = direct_dp_many;
  return be;
}

void ef_gpu_close(ef_backend* be) {
  if (!be) return;
  direct_be* d = (direct_be*)be->self;
  pgpu_index_destroy(d->ctx, d->idx);
  pgpu_destroy(d->ctx);
  ef_jobbuf_free(&d->jb);
  free(d->strings); free(d); free(be);
}
