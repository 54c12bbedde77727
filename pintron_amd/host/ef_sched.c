/* Batched execution of est-fact on the GPU.
 *
 * The per-EST algorithm (ef_compute_est_fact) is a sequential program whose pairing request and
 * dynamic programs depend on earlier results.  To keep one GPU busy with tens of thousands of
 * small, data-dependent requests, every input EST (the sequence and, when the strand is not fixed,
 * its reverse-complement sibling, tried only if the first fails: src/main-est-fact.c:249-291) runs
 * as a FIBRE: a request (or a group of independent requests) is recorded and the fibre yields.
 *
 *   workers   host threads; ESTs are dealt to them from a shared counter.  A worker owns a few
 *             LANES of fibres: it runs the runnable fibres of a lane, posts the lane's pending DP
 *             requests to the GPU service WITHOUT waiting and goes on with the next lane; it
 *             sleeps only when the lane it comes back to is still on the GPU.
 *   service   one or two threads that own the device queues: whatever has been posted is merged
 *             into ONE pgpu_dp_plan (one upload, one set of launches, one download), run, and the
 *             posters are woken to decode their slices of the shared result buffers.
 *   prefetch  the pairings of all ESTs at the configured parameters are computed in a few resident
 *             chunks (pgpu_pairing_plan_*) by a thread running beside the workers; only retries
 *             with a longer factor go through per-worker batches.
 *
 * The per-EST output text is kept in step-owned chunks and written in input order at the end, so
 * the files are byte-identical to a sequential run.
 */
#define _GNU_SOURCE
#include <malloc.h>
#include <pthread.h>
#include <sys/uio.h>
#include <errno.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <linux/futex.h>
#include <sys/resource.h>
#include <sched.h>
#include <dirent.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ucontext.h>
#include <sys/mman.h>

/* ---- fibre context switch --------------------------------------------------------------------------
 * swapcontext() saves and restores the signal mask with a system call on every switch; an EST
 * yields ~20 times, so on x86-64 the switch is done by hand: callee-saved registers on the own
 * stack, exchange the stack pointers.  Elsewhere ucontext is used. */
#if defined(__x86_64__) && !defined(EF_USE_UCONTEXT)
typedef struct { void* sp; } ef_ctx;
void ef_ctx_switch(void** save_sp, void* new_sp);
void ef_ctx_entry(void);
__asm__(
  ".text\n"
  ".globl ef_ctx_switch\n.type ef_ctx_switch,@function\n"
  "ef_ctx_switch:\n"
  "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
  "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
  "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
  "  ret\n"
  ".size ef_ctx_switch,.-ef_ctx_switch\n"
  ".globl ef_ctx_entry\n.type ef_ctx_entry,@function\n"
  "ef_ctx_entry:\n"                     /* first activation: r12 = argument, r13 = function */
  "  movq %r12, %rdi\n  callq *%r13\n  ud2\n"
  ".size ef_ctx_entry,.-ef_ctx_entry\n");
static inline void ctx_switch(ef_ctx* from, ef_ctx* to) { ef_ctx_switch(&from->sp, to->sp); }
/* fn(arg) starts on the given stack at the first switch to c; fn must not return */
static void ctx_make(ef_ctx* c, char* stack, size_t size, void (*fn)(void*), void* arg) {
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  void** S = (void**)(top - 16);         /* 16-byte aligned: the call in ef_ctx_entry leaves rsp = 8 mod 16 */
  S[-1] = (void*)ef_ctx_entry;           /* return address of the first switch */
  S[-2] = NULL;                          /* rbp */
  S[-3] = NULL;                          /* rbx */
  S[-4] = arg;                           /* r12 */
  S[-5] = (void*)fn;                     /* r13 */
  S[-6] = NULL;                          /* r14 */
  S[-7] = NULL;                          /* r15 */
  c->sp = (void*)(S - 7);
}
#else
typedef struct { ucontext_t uc; } ef_ctx;
static inline void ctx_switch(ef_ctx* from, ef_ctx* to) { swapcontext(&from->uc, &to->uc); }
static void ctx_tramp(unsigned fh, unsigned fl, unsigned ah, unsigned al) {
  void (*fn)(void*) = (void (*)(void*))(((uintptr_t)fh << 32) | (uintptr_t)fl);
  fn((void*)(((uintptr_t)ah << 32) | (uintptr_t)al));
}
static void ctx_make(ef_ctx* c, char* stack, size_t size, void (*fn)(void*), void* arg) {
  getcontext(&c->uc);
  c->uc.uc_stack.ss_sp = stack; c->uc.uc_stack.ss_size = size; c->uc.uc_link = NULL;
  const uintptr_t f = (uintptr_t)fn, a = (uintptr_t)arg;
  makecontext(&c->uc, (void (*)(void))ctx_tramp, 4, (unsigned)(f >> 32), (unsigned)(f & 0xffffffffu), (unsigned)(a >> 32), (unsigned)(a & 0xffffffffu));
}
#endif
/* ThreadSanitizer has to be told about the switches (tools/tsan_hostcheck.py); nothing of this is
 * compiled otherwise */
#if defined(__SANITIZE_THREAD__)
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
#define EF_TSAN 1
#else
#define EF_TSAN 0
#endif
#include <unistd.h>

#include "estfact.h"
#include "ef_gpu.h"
#include "ef_sched.h"

enum { F_RUNNABLE, F_WAIT_DP, F_WAIT_PAIR, F_DONE };

#if EF_TSAN
static inline void tsan_to(void* fiber) { __tsan_switch_to_fiber(fiber, 0); }
#else
static inline void tsan_to(void* fiber) { (void)fiber; }
#endif

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

struct worker;
#define EF_N_OUT 7            /* six output files + the packed factorization records */

typedef struct fiber {
  ef_ctx ctx;
  void* tsan;                /* ThreadSanitizer's handle of this fibre (EF_TSAN builds) */
  char* stack; bool guarded;
  struct worker* w;
  int state;
  size_t unit;
  size_t cur_entry;          /* entry of the prepared list being factorized */
  int lane;
  int phase;                 /* PINTRON_PROFILE: what the per-EST code is doing (estfact.h: EFP_*) */
  /* pending request */
  const ef_dp_req* reqs; ef_dp_res* ress; size_t nreq;   /* nreq independent DP requests */
  ef_dp_req req1; int rc;
  const char* pat; size_t pat_len; unsigned pat_L; double pat_rate; ef_triple** pat_out; size_t* pat_n;
  ef_backend be;
  struct fiber* pool_next;   /* free fibres (struct + stack + sink blocks) are kept for the next EST */
  ef_sink out[EF_N_OUT];     /* text of the unit being processed */
} fiber;

static const char FIBER_SENTINEL[16] = "pintron-fibre-s";

static inline void fiber_prefetch(const fiber* f) {
#if defined(__x86_64__) && !defined(EF_USE_UCONTEXT)
  const char* sp = (const char*)f->ctx.sp;
  for (int k = 0; k < 12; ++k) __builtin_prefetch(sp + 64 * k, 0, 3);
#endif
  __builtin_prefetch(f->ress, 0, 3);
  __builtin_prefetch(f->reqs, 0, 3);
}

/* Fibre stacks are their own mappings with an inaccessible page below the lowest address: a stack
 * that overflows (the embedding enumeration recurses as deep as the MEG is long) faults on the
 * guard page instead of writing over a neighbouring heap block.  When the mapping cannot be had
 * (map count exhausted by a very large PINTRON_FIBERS) the stack is a heap block and only the
 * sentinel at its low end tells. */
#define EF_GUARD_BYTES 4096u
static char* stack_alloc(size_t size, bool* guarded) {
  void* m = mmap(NULL, size + EF_GUARD_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
  if (m != MAP_FAILED && mprotect(m, EF_GUARD_BYTES, PROT_NONE) == 0) { *guarded = true; return (char*)m + EF_GUARD_BYTES; }
  if (m != MAP_FAILED) munmap(m, size + EF_GUARD_BYTES);
  *guarded = false;
  return (char*)malloc(size);
}
static void stack_free(char* stack, size_t size, bool guarded) {
  if (!stack) return;
  if (guarded) munmap(stack - EF_GUARD_BYTES, size + EF_GUARD_BYTES); else free(stack);
}

/* one input EST: entry `first` of the prepared list, plus the sibling at first+1 if any */
typedef struct {
  size_t first; bool has_sibling;
  char* buf[EF_N_OUT]; size_t len[EF_N_OUT];   /* raw, processed-ests, megs, processed-megs, megs-info,
                                                * meg-edges, packed factorization records */
} unit;

/* ---- GPU service: one thread owns the device queue ------------------------------------------------
 * Workers do not talk to the GPU themselves.  A worker whose fibres are all blocked posts its
 * pending DP requests and SLEEPS (condition variable); the service thread takes everything that
 * has been posted so far, merges it into ONE plan (one upload, one set of kernel launches, one
 * download), and wakes the posters, which decode their slice of the shared result buffers in
 * parallel.  With more workers than cores the CPUs stay busy with host logic while other workers
 * sleep, and the GPU sees few, large batches instead of many small ones. */
typedef struct merged_batch {
  pgpu_dp_result* results; char* strings;
  int refs;                              /* posters that still have to decode their slice */
} merged_batch;

typedef struct dp_request {
  const pgpu_dp_job* jobs; size_t n; const char* arena; size_t arena_len;
  merged_batch* batch; size_t base;      /* filled by the service */
  int rc;
  uint32_t done;                         /* 0 -> 1 by the service, written last (release) */
  uint32_t* wake;                        /* the posting worker's wake word: bumped and woken when this request is done,
                                            so a finished batch wakes the posters of ITS requests (not every worker that
                                            waits for some batch) and a worker with all lanes in flight resumes on
                                            whichever of its batches returns first */
  struct dp_request* next;
} dp_request;

static inline void request_publish(dp_request* rq) {
  uint32_t* wake = rq->wake;             /* the owner may reuse the request the moment `done` is set */
  __atomic_store_n(&rq->done, 1u, __ATOMIC_RELEASE);
  __atomic_add_fetch(wake, 1u, __ATOMIC_RELEASE);
  syscall(SYS_futex, wake, FUTEX_WAKE_PRIVATE, 1, NULL, NULL, 0);
}
/* sleep until the worker's wake word moves on from `seen` */
static inline void worker_sleep(uint32_t* wake, uint32_t seen) {
  syscall(SYS_futex, wake, FUTEX_WAIT_PRIVATE, seen, NULL, NULL, 0);
}

#define MAX_SERVICES 8
#define EF_MAX_WORKERS 256
typedef struct service_thread {          /* one submitter: own context (stream, device buffers) */
  pthread_t thread;
  struct service* sv;
  pgpu_ctx* ctx;
  ef_sched_stats stats;                  /* batches, jobs, kernel timings */
  double* iv; size_t n_iv, cap_iv;       /* [start, end) of every timed kernel launch on the device's time line (ms) */
  double phase_s[6];                     /* idle, idle+merge, create, launch, sync, fetch+stats */
} service_thread;

typedef struct service {
  pthread_mutex_t mu;
  pthread_cond_t posted, finished;
  dp_request *head, *tail;
  bool stop;
  struct shared* sh;
  int n_threads;                         /* PINTRON_SERVICES: batches of different submitters overlap on the GPU */
  long coalesce_us;                      /* PINTRON_COALESCE_US: wait that long for more requests before merging */
  service_thread threads[MAX_SERVICES];
} service;

#define PRE_CHUNKS 16                 /* at most; PINTRON_PRE_CHUNKS ranges are used (default 10) */
typedef struct shared {
  ef_inputs* in;
  pgpu_index* idx;
  unit* units; size_t n_units; size_t units_cap;
  size_t next_unit;                    /* dealt out with an atomic add (start_fiber) */
  pthread_mutex_t mu;
  size_t max_fibers, stack_size;
  /* pairings of every list entry at the configured (min_factor_len, rate), computed in ONE
   * resident batch before the fibres start; retries with a longer factor go through batches */
  /* ... in PRE_CHUNKS ranges of entries, computed by a prefetch thread while the workers already
   * run on the finished ranges: chunk c covers entries [pre_lo[c], pre_lo[c+1]) */
  int n_pre;                                  /* 0 = no prefetch */
  size_t pre_lo[PRE_CHUNKS + 1];
  pgpu_pairing* pre_tri[PRE_CHUNKS]; uint64_t* pre_first[PRE_CHUNKS];
  /* ... or, with the MEG stage on the device (the default), the finished graphs instead of the
   * pairings: records in page-locked buffers that live as long as the session */
  bool use_meg;
  pgpu_meg_params meg_prm;
  unsigned char* pre_meg[PRE_CHUNKS]; size_t pre_meg_cap[PRE_CHUNKS]; uint64_t* pre_meg_first[PRE_CHUNKS];
  /* page-locked memory is slow to get (0.25 s per GB): the records and offsets of all chunks are cut from ONE
   * allocation sized after the first chunk (a chunk that does not fit gets a block of its own), and the
   * patterns of every chunk go up through one staging buffer */
  unsigned char* pre_slab; size_t pre_slab_cap, pre_slab_used;
  bool pre_own[PRE_CHUNKS], pre_first_own[PRE_CHUNKS];
  char* up_stage; size_t up_stage_cap;
  size_t ready_entries;                       /* entries below this have their pairings (written under mu with a release store; read without the lock) */
  pthread_cond_t ready_cv;
  bool kernel_timing;
  size_t gen_len;
  int n_lanes;
  service svc;
  int failed;
  ef_sched_stats stats;
  /* fibres (256 KB stacks) are recycled: within a worker through its own free list, across steps
   * through this one (under mu) */
  fiber* fiber_pool;
  /* ... and what a worker had at the end of a step is what worker of the same number starts the next step with:
   * sixteen workers drawing their first thousand fibres one by one from the shared list queue up on its lock
   * (13 - 20 ms of every warm step, PINTRON_PROFILE "sched:start") */
  fiber* worker_pool[EF_MAX_WORKERS];
  /* output text of the units lives in large chunks owned by the step (freed together) */
  struct out_chunk* chunks;              /* under mu */
  struct out_chunk* spare_chunks;        /* chunks of the previous step, reused (their pages stay mapped) */
  unsigned long long prof_cyc[EFP_N], prof_susp[EFP_N], prof_jobs[EFP_N];   /* PINTRON_PROFILE, summed over the workers (under mu) */
  unsigned long long prof_ahead[5];
  double prof_t0; double prof_sleep_bins[64];      /* when in the step the workers slept (5 ms bins, seconds summed over the workers) */
} shared;

typedef struct out_chunk { struct out_chunk* next; size_t cap, used; char data[]; } out_chunk;

/* A worker keeps several independent sets of fibres ("lanes"): while the requests of one lane are
 * with the GPU service, the fibres of the other lanes run on the CPU, so the worker only sleeps
 * when every lane is waiting. */
#define MAX_LANES 16
typedef struct lane {
  fiber** fibers; size_t n_fibers;
  ef_jobbuf jb;
  dp_request rq; bool posted;    /* requests handed to the GPU service, not yet collected */
  fiber** inflight; size_t n_inflight;
} lane;

typedef struct worker {
  shared* sh;
  size_t index;                          /* of this worker among the step's workers */
  double sleep_bins[64];
  size_t suspensions;                    /* times a fibre of this worker gave up the thread for an answer */
  ef_ctx sched;
  void* tsan_sched;          /* the worker thread's own context, for ThreadSanitizer */
  fiber* free_fibers;
  out_chunk* chunk;                      /* current output chunk of this worker */
  lane lanes[MAX_LANES];
  uint32_t wake;                         /* see dp_request */
  pgpu_ctx* ctx;                         /* for pairing retries only; created on first use */
  size_t held_unit; bool holding;        /* a unit this worker has drawn whose pairings are still on their way (start_fiber) */
  ef_sched_stats stats;
} worker;

static void kstat_add(ef_sched_stats* st, const ef_kernel_stat* k) {
  for (int i = 0; i < st->n_kernels; ++i)
    if (!strcmp(st->kernels[i].name, k->name)) {
      st->kernels[i].ms += k->ms; st->kernels[i].launches += k->launches; st->kernels[i].jobs += k->jobs;
      st->kernels[i].cells += k->cells; st->kernels[i].algo_bytes += k->algo_bytes;
      return;
    }
  if (st->n_kernels < EF_MAX_KERNELS) st->kernels[st->n_kernels++] = *k;
}

/* ---- fibre side ---------------------------------------------------------------------------------- */
static int fiber_dp_many(void* self, const ef_dp_req* reqs, ef_dp_res* res, size_t n) {
  fiber* f = (fiber*)self;
  if (n == 0) return 0;
  f->reqs = reqs; f->ress = res; f->nreq = n; f->state = F_WAIT_DP;
  ++f->w->suspensions;
  if (ef_prof_on) { ef_prof.susp[f->phase]++; ef_prof.jobs[f->phase] += n; }
  if (EF_TSAN) tsan_to(f->w->tsan_sched);
  ctx_switch(&f->ctx, &f->w->sched);
  return f->rc;
}

static int fiber_dp(void* self, const ef_dp_req* q, ef_dp_res* res) {
  fiber* f = (fiber*)self;
  f->req1 = *q;
  return fiber_dp_many(self, &f->req1, res, 1);
}

static int fiber_pairings(void* self, const char* pattern, size_t m, unsigned L, double rate, ef_triple** out, size_t* n) {
  fiber* f = (fiber*)self;
  const shared* sh = f->w->sh;
  int c = 0;
  if (sh->n_pre) while (f->cur_entry >= sh->pre_lo[c + 1]) ++c;
  if (sh->n_pre && sh->pre_tri[c] && L == sh->in->cfg.min_factor_len && rate == sh->in->cfg.min_string_depth_rate &&
      pattern == sh->in->list[f->cur_entry]->seq) {
    const size_t e = f->cur_entry - sh->pre_lo[c];
    const uint64_t a = sh->pre_first[c][e], b = sh->pre_first[c][e + 1];
    ef_triple* t = (ef_triple*)malloc((size_t)(b - a + 1) * sizeof(ef_triple));
    memcpy(t, sh->pre_tri[c] + a, (size_t)(b - a) * sizeof(ef_triple));
    *out = t; *n = (size_t)(b - a);
    return 0;
  }
  f->pat = pattern; f->pat_len = m; f->pat_L = L; f->pat_rate = rate; f->pat_out = out; f->pat_n = n;
  f->state = F_WAIT_PAIR;
  if (EF_TSAN) tsan_to(f->w->tsan_sched);
  ctx_switch(&f->ctx, &f->w->sched);
  return f->rc;
}

/* the device-built MEG of the entry being factorized, at the configured parameters */
static const void* fiber_meg(void* self, const char* pattern, size_t m, const ef_config* cfg) {
  (void)m;
  fiber* f = (fiber*)self;
  const shared* sh = f->w->sh;
  if (!sh->use_meg || !sh->n_pre || cfg->min_factor_len != sh->in->cfg.min_factor_len ||
      cfg->min_string_depth_rate != sh->in->cfg.min_string_depth_rate || pattern != sh->in->list[f->cur_entry]->seq)
    return NULL;
  int c = 0;
  while (f->cur_entry >= sh->pre_lo[c + 1]) ++c;
  if (!sh->pre_meg[c]) return NULL;
  return sh->pre_meg[c] + sh->pre_meg_first[c][f->cur_entry - sh->pre_lo[c]];
}

/* room for n bytes of output text in the worker's current chunk */
static char* out_alloc(worker* w, size_t n) {
  out_chunk* c = w->chunk;
  if (!c || c->used + n > c->cap) {
    const size_t cap = n > (1u << 20) ? n : (1u << 20);
    shared* sh = w->sh;
    c = NULL;
    pthread_mutex_lock(&sh->mu);
    if (sh->spare_chunks && sh->spare_chunks->cap >= n) { c = sh->spare_chunks; sh->spare_chunks = c->next; }
    pthread_mutex_unlock(&sh->mu);
    if (!c) { c = (out_chunk*)malloc(sizeof(out_chunk) + cap); c->cap = cap; }
    c->used = 0;
    pthread_mutex_lock(&sh->mu);
    c->next = sh->chunks; sh->chunks = c;
    pthread_mutex_unlock(&sh->mu);
    w->chunk = c;
  }
  char* r = c->data + c->used;
  c->used += n;
  return r;
}

static void fiber_main(void* arg) {
  fiber* f = (fiber*)arg;
  shared* sh = f->w->sh;
  unit* u = &sh->units[f->unit];
  /* the text of the unit grows in the fibre's own memory sinks (their blocks are kept when the
   * fibre is recycled) and is copied into the step's chunks when the unit is done */
  ef_sink* fs = f->out;
  for (int k = 0; k < EF_N_OUT; ++k) { fs[k].f = NULL; fs[k].len = 0; }
  ef_side_files side = { &fs[2], &fs[3], &fs[4], &fs[5] };
  const ef_inputs* in = sh->in;
  for (size_t k = u->first; k <= u->first + (u->has_sibling ? 1 : 0); ++k) {
    f->cur_entry = k;
    ef_est* fe = ef_compute_est_fact(in->gen, in->list[k], &f->be, &in->cfg, &side);
    const bool aligned = !efl_empty(fe->factorizations);
    ef_phase(EFP_OUTPUT);
    if (aligned) {
      ef_write_multifasta_output(in->gen, fe, &fs[0], in->cfg.retain_externals);
      ef_write_factorization_records(in->gen, fe, &fs[6], in->cfg.retain_externals, (uint32_t)f->unit);
      ef_write_single_est_info(&fs[1], fe->info);
    }
    ef_phase(EFP_FREE);
    ef_est_free(fe);
    ef_phase(EFP_OTHER);
    if (aligned) break;
  }
  ef_phase(EFP_OUTPUT);
  for (int k = 0; k < EF_N_OUT; ++k) {
    u->len[k] = fs[k].len;
    u->buf[k] = fs[k].len ? out_alloc(f->w, fs[k].len) : NULL;
    if (fs[k].len) memcpy(u->buf[k], fs[k].mem, fs[k].len);
  }
  f->state = F_DONE;
  if (EF_TSAN) tsan_to(f->w->tsan_sched);
  ctx_switch(&f->ctx, &f->w->sched);
}

/* ---- worker side --------------------------------------------------------------------------------- */
static inline bool unit_ready(const shared* sh, size_t u) {
  if (!sh->n_pre) return true;
  const size_t last = sh->units[u].first + (sh->units[u].has_sibling ? 1 : 0);
  return last < __atomic_load_n(&sh->ready_entries, __ATOMIC_ACQUIRE);
}
static void wait_unit_ready(shared* sh, size_t u) {
  const size_t last = sh->units[u].first + (sh->units[u].has_sibling ? 1 : 0);
  const int was = ef_phase(EFP_WAIT_PREFETCH);
  pthread_mutex_lock(&sh->mu);
  while (last >= sh->ready_entries && !sh->failed) pthread_cond_wait(&sh->ready_cv, &sh->mu);
  pthread_mutex_unlock(&sh->mu);
  ef_phase(was);
}
/* the worker has fibres to run or batches to wait for */
static inline bool worker_has_work(const worker* w) {
  for (int k = 0; k < w->sh->n_lanes; ++k) if (w->lanes[k].n_fibers > 0 || w->lanes[k].posted) return true;
  return false;
}

enum { START_NONE = 0, START_OK = 1, START_LATER = 2 };
/* START_LATER: the next unit's pairings are still on their way and the worker has other fibres to run -- the unit
 * stays with the worker (held_unit) and is started by a later call.  (Until round 4 the worker slept here with
 * runnable fibres in its other lanes: harmless with ten equal ranges, where only the first is waited for, and the
 * reason small first ranges lost.) */
static int start_fiber(worker* w, int li) {
  shared* sh = w->sh;
  lane* ln = &w->lanes[li];
  fiber* f = w->free_fibers;
  /* The units are dealt out with one atomic add: a step over a C5 share hands out 1.5 million units per
   * second to sixteen workers, and a mutex taken per unit turns into a convoy now and then (the same run
   * then needs 250 ms instead of 170).  The lock is only taken to wait for the prefetch stage and to draw
   * from the shared fibre pool (first step). */
  size_t u;
  if (w->holding) u = w->held_unit;
  else {
    u = __atomic_fetch_add(&sh->next_unit, 1, __ATOMIC_RELAXED);
    if (u >= sh->n_units) return START_NONE;
  }
  if (!unit_ready(sh, u)) {                    /* the pairings of this unit are still on their way */
    if (worker_has_work(w)) { w->held_unit = u; w->holding = true; return START_LATER; }
    wait_unit_ready(sh, u);
  }
  w->holding = false;
  if (__atomic_load_n(&sh->failed, __ATOMIC_RELAXED)) return START_NONE;
  if (f) w->free_fibers = f->pool_next;
  else if (__atomic_load_n(&sh->fiber_pool, __ATOMIC_RELAXED)) {
    pthread_mutex_lock(&sh->mu);                 /* a few dozen at a time: the lock is shared by all workers */
    if (sh->fiber_pool) {
      f = sh->fiber_pool;
      fiber* last = f;
      for (int k = 0; k < 32 && last->pool_next; ++k) last = last->pool_next;
      sh->fiber_pool = last->pool_next;
      last->pool_next = NULL;
      w->free_fibers = f->pool_next; f->pool_next = NULL;
    }
    pthread_mutex_unlock(&sh->mu);
  }
  if (f) {                       /* recycled: keep the stack and the sink blocks */
    char* st = f->stack;
    const bool gd = f->guarded;
    void* ts = f->tsan;
    ef_sink keep[EF_N_OUT];
    memcpy(keep, f->out, sizeof keep);
    memset(f, 0, sizeof(fiber));
    f->stack = st; f->guarded = gd; f->tsan = ts;
    memcpy(f->out, keep, sizeof keep);
  }
  else { f = (fiber*)calloc(1, sizeof(fiber)); f->stack = stack_alloc(sh->stack_size, &f->guarded); }
  f->w = w; f->unit = u; f->state = F_RUNNABLE; f->lane = li;
  f->be.self = f; f->be.pairings = fiber_pairings; f->be.dp = fiber_dp; f->be.dp_many = fiber_dp_many; f->be.meg = fiber_meg;
  /* besides the guard page: a sentinel at the low end, checked when the fibre is done */
  memcpy(f->stack, FIBER_SENTINEL, sizeof FIBER_SENTINEL);
#if EF_TSAN
  if (f->tsan) __tsan_destroy_fiber(f->tsan);
  f->tsan = __tsan_create_fiber(0);
  w->tsan_sched = __tsan_get_current_fiber();
#endif
  ctx_make(&f->ctx, f->stack, sh->stack_size, fiber_main, f);
  ln->fibers[ln->n_fibers++] = f;
  return START_OK;
}

static int submit_pairings(worker* w, lane* ln) {
  /* group the waiting fibres by (L, rate): a retry with a longer min_factor_len runs separately */
  fiber** wait = (fiber**)malloc((ln->n_fibers + 1) * sizeof(fiber*));
  size_t nw = 0;
  for (size_t i = 0; i < ln->n_fibers; ++i) if (ln->fibers[i]->state == F_WAIT_PAIR) wait[nw++] = ln->fibers[i];
  int rc = 0;
  while (nw > 0 && rc == 0) {
    const unsigned L = wait[0]->pat_L; const double rate = wait[0]->pat_rate;
    size_t ng = 0, total = 0;
    for (size_t i = 0; i < nw; ++i) if (wait[i]->pat_L == L && wait[i]->pat_rate == rate) { ++ng; total += wait[i]->pat_len; }
    char* blob = (char*)malloc(total + 1);
    uint64_t* off = (uint64_t*)malloc((ng + 1) * sizeof(uint64_t));
    fiber** grp = (fiber**)malloc(ng * sizeof(fiber*));
    size_t g = 0, pos = 0, rest = 0;
    for (size_t i = 0; i < nw; ++i) {
      fiber* f = wait[i];
      if (f->pat_L == L && f->pat_rate == rate) { off[g] = pos; memcpy(blob + pos, f->pat, f->pat_len); pos += f->pat_len; grp[g++] = f; }
      else wait[rest++] = f;
    }
    off[ng] = pos;
    pgpu_pairing_plan* plan = NULL;
    pgpu_pairing_params prm = { L, 0, rate };
    if (!w->ctx && pgpu_init(ef_gpu_device_from_env(), &w->ctx) != PGPU_OK) { free(blob); free(off); free(grp); free(wait); return 1; }
    rc = pgpu_pairing_plan_create(w->ctx, w->sh->idx, blob, off, ng, &plan);
    if (rc == PGPU_OK) rc = pgpu_pairing_plan_run(w->ctx, plan, &prm);
    if (rc == PGPU_OK) {
      const size_t cnt = (size_t)pgpu_pairing_plan_count(plan);
      pgpu_pairing* out = (pgpu_pairing*)malloc((cnt + 1) * sizeof(pgpu_pairing));
      uint64_t* first = (uint64_t*)malloc((ng + 1) * sizeof(uint64_t));
      rc = pgpu_pairing_plan_fetch(w->ctx, plan, out, cnt, first);
      if (rc == PGPU_OK) {
        for (size_t i = 0; i < ng; ++i) {
          const size_t n = (size_t)(first[i + 1] - first[i]);
          ef_triple* t = (ef_triple*)malloc((n + 1) * sizeof(ef_triple));
          memcpy(t, out + first[i], n * sizeof(ef_triple));
          *grp[i]->pat_out = t; *grp[i]->pat_n = n;
          grp[i]->rc = 0; grp[i]->state = F_RUNNABLE;
        }
        w->stats.pairing_batches++; w->stats.pairing_requests += ng;
      }
      free(out); free(first);
    }
    if (plan) pgpu_pairing_plan_destroy(w->ctx, plan);
    if (rc != PGPU_OK) fprintf(stderr, "* FATAL pairing batch failed: %s\n", pgpu_last_error(w->ctx));
    free(blob); free(off); free(grp);
    nw = rest;
  }
  free(wait);
  return rc;
}

static void kstat_add(ef_sched_stats* st, const ef_kernel_stat* k);

/* service thread: merge everything posted, run it as one plan, publish the results */
static void* service_main(void* arg) {
  pthread_setname_np(pthread_self(), "ef-gpu-service");
  /* the library waits for a batch in short sleeps (PGPU_WAIT, microseconds): with the default timer
   * slack of 50 us a 20 us sleep takes 70 */
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
  service_thread* me = (service_thread*)arg;
  service* sv = me->sv;
  shared* sh = sv->sh;
  pgpu_dp_part* parts = NULL; size_t parts_cap = 0;
  for (;;) {
    pthread_mutex_lock(&sv->mu);
    const double t_idle = now_s();
    while (!sv->head && !sv->stop) pthread_cond_wait(&sv->posted, &sv->mu);
    me->phase_s[0] += now_s() - t_idle;
    if (sv->coalesce_us > 0 && sv->head && !sv->stop) {
      /* a short wait lets the lanes that are about to post join this batch: fewer, larger batches
       * (every launch of a batch is latency-bound, so its cost hardly grows with the job count) */
      pthread_mutex_unlock(&sv->mu);
      struct timespec ts = { 0, sv->coalesce_us * 1000L };
      nanosleep(&ts, NULL);
      pthread_mutex_lock(&sv->mu);
    }
    dp_request* list = sv->head;
    sv->head = sv->tail = NULL;
    const bool stop = sv->stop;
    pthread_mutex_unlock(&sv->mu);
    if (!list) { if (stop) break; else continue; }
    /* the requests of the posters become the parts of one plan: the library gathers their jobs and
     * operand arenas straight into its pinned upload image */
    size_t nj = 0; int nreq = 0;
    for (dp_request* r = list; r; r = r->next) ++nreq;
    if ((size_t)nreq > parts_cap) { parts_cap = (size_t)nreq * 2; parts = (pgpu_dp_part*)realloc(parts, parts_cap * sizeof(pgpu_dp_part)); }
    {
      size_t q = 0;
      for (dp_request* r = list; r; r = r->next, ++q) {
        r->base = nj;
        parts[q].jobs = r->jobs; parts[q].n_jobs = r->n; parts[q].arena = r->arena; parts[q].arena_len = r->arena_len;
        nj += r->n;
      }
    }
    merged_batch* mb = (merged_batch*)calloc(1, sizeof(merged_batch));
    mb->refs = nreq;
    mb->results = (pgpu_dp_result*)malloc((nj + 1) * sizeof(pgpu_dp_result));
    pgpu_dp_plan* plan = NULL;
    const double t_a = now_s();
    me->phase_s[1] += t_a - t_idle - 0;        /* (includes the idle wait; corrected below) */
    pgpu_range_push("dp batch (create, launch, sync, fetch)");
    int rc = pgpu_dp_plan_create_parts(me->ctx, sh->idx, parts, (size_t)nreq, &plan);
    const double t_b = now_s();
    if (rc == PGPU_OK) rc = pgpu_dp_plan_launch(me->ctx, plan);
    const double t_c = now_s();
    if (rc == PGPU_OK) rc = pgpu_dp_plan_sync(me->ctx, plan);
    const double t_d = now_s();
    me->phase_s[2] += t_b - t_a; me->phase_s[3] += t_c - t_b; me->phase_s[4] += t_d - t_c;
    if (rc == PGPU_OK) {
      const size_t sb = pgpu_dp_plan_string_bytes(plan);
      mb->strings = (char*)malloc(sb + 16);
      rc = pgpu_dp_plan_fetch(me->ctx, plan, mb->results, mb->strings, sb + 16);
    }
    if (rc == PGPU_OK && sh->kernel_timing) {
      const int ng = pgpu_dp_plan_n_groups(plan);
      for (int g = 0; g < ng; ++g) {
        pgpu_group_info gi;
        if (pgpu_dp_plan_group_info(plan, g, &gi) != PGPU_OK || gi.jobs == 0) continue;   /* nothing launched for it */
        ef_kernel_stat ks; memset(&ks, 0, sizeof ks);
        snprintf(ks.name, sizeof ks.name, "%s", gi.name);
        ks.ms = gi.ms; ks.launches = 1; ks.jobs = gi.jobs; ks.cells = gi.cells; ks.algo_bytes = gi.algo_bytes;
        kstat_add(&me->stats, &ks);
        if (gi.t0_ms >= 0 && gi.ms > 0) {
          if (me->n_iv + 2 > me->cap_iv) { me->cap_iv = me->cap_iv ? me->cap_iv * 2 : 4096; me->iv = (double*)realloc(me->iv, me->cap_iv * sizeof(double)); }
          me->iv[me->n_iv++] = gi.t0_ms; me->iv[me->n_iv++] = gi.t0_ms + gi.ms;
        }
      }
    }
    if (plan) pgpu_dp_plan_destroy(me->ctx, plan);
    pgpu_range_pop();
    me->phase_s[5] += now_s() - t_d;
    if (rc != PGPU_OK) fprintf(stderr, "* FATAL DP batch failed: %s\n", pgpu_last_error(me->ctx));
    me->stats.dp_batches++; me->stats.dp_jobs += nj;
    /* `done` is what the poster sleeps on and what its lane choice polls: written last (release); the
     * request may be reused by its owner the moment it is published, so `next` is read before */
    for (dp_request* r = list; r;) { dp_request* nx = r->next; r->batch = mb; r->rc = rc; request_publish(r); r = nx; }
  }
  free(parts);
  return NULL;
}

/* post the pending DP requests of a lane to the GPU service (does not wait) */
static int launch_dp(worker* w, lane* ln) {
  shared* sh = w->sh;
  service* sv = &sh->svc;
  const char* gen = sh->in->gen->seq;
  ef_jobbuf_reset(&ln->jb);
  ln->n_inflight = 0;
  for (size_t i = 0; i < ln->n_fibers; ++i) {
    fiber* f = ln->fibers[i];
    if (f->state != F_WAIT_DP) continue;
    for (size_t k = 0; k < f->nreq; ++k) ef_jobbuf_add(&ln->jb, &f->reqs[k], gen, sh->gen_len);
    ln->inflight[ln->n_inflight++] = f;
  }
  if (ln->n_inflight == 0) return 0;
  dp_request* rq = &ln->rq;
  memset(rq, 0, sizeof *rq);
  rq->wake = &w->wake;
  rq->jobs = ln->jb.jobs; rq->n = ln->jb.n; rq->arena = ln->jb.arena; rq->arena_len = ln->jb.arena_len;
  pthread_mutex_lock(&sv->mu);
  if (sv->tail) sv->tail->next = rq; else sv->head = rq;
  sv->tail = rq;
  pthread_cond_signal(&sv->posted);
  pthread_mutex_unlock(&sv->mu);
  ln->posted = true;
  return 0;
}

/* sleep until the lane's posted requests are back, then decode this lane's slice */
static int collect_dp(worker* w, lane* ln) {
  if (!ln->posted) return 0;
  dp_request* rq = &ln->rq;
  for (;;) {
    const uint32_t seen = __atomic_load_n(&w->wake, __ATOMIC_ACQUIRE);
    if (__atomic_load_n(&rq->done, __ATOMIC_ACQUIRE)) break;
    ef_phase(EFP_SLEEP);
    worker_sleep(&w->wake, seen);
    ef_phase(EFP_SCHED_COLLECT);
  }
  ln->posted = false;
  const int rc = rq->rc;
  const bool tracing = ef_dp_trace_enabled() != 0;
  if (rc == PGPU_OK) {
    size_t j = rq->base;
    for (size_t i = 0; i < ln->n_inflight; ++i) {
      fiber* f = ln->inflight[i];
      f->rc = 0;
      for (size_t k = 0; k < f->nreq; ++k, ++j) {
        const int drc = ef_decode_result(f->reqs[k].kind, &rq->batch->results[j], rq->batch->strings, &f->ress[k]);
        if (drc != 0) {
          fprintf(stderr, "* FATAL DP job of kind %d (%zu x %zu) exceeds the device limits\n", f->reqs[k].kind, f->reqs[k].la, f->reqs[k].lb);
          f->rc = drc;
        }
        else if (tracing) ef_dp_trace(&f->reqs[k], &f->ress[k], (uint32_t)f->unit);
      }
      f->state = F_RUNNABLE;
    }
  }
  const bool last = __atomic_sub_fetch(&rq->batch->refs, 1, __ATOMIC_ACQ_REL) == 0;
  if (last) { free(rq->batch->results); free(rq->batch->strings); free(rq->batch); }
  ln->n_inflight = 0;
  return rc;
}

static void* worker_main(void* arg) {
  pthread_setname_np(pthread_self(), "ef-worker");
  worker* w = (worker*)arg;
  shared* sh = w->sh;
  const int n_lanes = sh->n_lanes;
  const size_t per_lane = sh->max_fibers / n_lanes ? sh->max_fibers / n_lanes : 1;
  for (int li = 0; li < n_lanes; ++li) {
    lane* ln = &w->lanes[li];
    ln->fibers = (fiber**)malloc(per_lane * sizeof(fiber*));
    ln->inflight = (fiber**)malloc(per_lane * sizeof(fiber*));
    ef_jobbuf_init(&ln->jb);
  }
  if (ef_prof_on) { memset(&ef_prof, 0, sizeof ef_prof); ef_prof.dummy = EFP_SCHED; ef_prof.last = ef_prof_now(); }
  if (w->index < EF_MAX_WORKERS) { w->free_fibers = sh->worker_pool[w->index]; sh->worker_pool[w->index] = NULL; }
  bool more = true;
  int cursor = 0;
  const bool prefetch_on = !getenv("PINTRON_NO_FIBER_PREFETCH");
  while (!sh->failed) {
    /* next lane: the first one (round robin) that is not waiting for the GPU -- nothing posted, or
     * its batch is back; when every lane is in flight, sleep on the one posted longest ago */
    int li = -1, first_posted = -1;
    const bool can_start = more && (!w->holding || unit_ready(sh, w->held_unit));
    for (int k = 0; k < n_lanes; ++k) {
      const int idx = (cursor + k) % n_lanes;
      lane* c = &w->lanes[idx];
      if (c->posted) {
        if (__atomic_load_n(&c->rq.done, __ATOMIC_ACQUIRE)) { li = idx; break; }
        if (first_posted < 0) first_posted = idx;
      } else if (c->n_fibers > 0 || can_start) { li = idx; break; }
    }
    if (li < 0 && first_posted < 0 && more && w->holding) {       /* nothing but the unit that is not ready yet */
      wait_unit_ready(sh, w->held_unit);
      if (sh->failed) break;
      continue;
    }
    if (li < 0 && first_posted >= 0) {
      /* everything that has work is on the GPU: sleep until any of this worker's batches is back
       * (the wake word is read before the lanes are looked at again, so a completion in between is
       * not slept through) */
      const double tw = now_s();
      ef_phase(EFP_SLEEP);
      for (;;) {
        const uint32_t seen = __atomic_load_n(&w->wake, __ATOMIC_ACQUIRE);
        for (int k = 0; k < n_lanes && li < 0; ++k) {
          const int idx = (cursor + k) % n_lanes;
          if (w->lanes[idx].posted && __atomic_load_n(&w->lanes[idx].rq.done, __ATOMIC_ACQUIRE)) li = idx;
        }
        if (li >= 0 || sh->failed) break;
        worker_sleep(&w->wake, seen);
      }
      w->stats.dp_s += now_s() - tw;
      if (ef_prof_on) {
        const double t1 = now_s();
        int b = (int)((tw - sh->prof_t0) / 0.005); if (b < 0) b = 0; if (b > 63) b = 63;
        w->sleep_bins[b] += t1 - tw;
      }
      ef_phase(EFP_SCHED);
      if (li < 0) break;
    }
    if (li < 0) break;                   /* no fibres, no ESTs left, nothing in flight */
    cursor = (li + 1) % n_lanes;
    lane* ln = &w->lanes[li];
    double t0 = now_s();
    ef_phase(EFP_SCHED_COLLECT);
    if (collect_dp(w, ln) != 0) { sh->failed = 1; break; }
    w->stats.dp_s += now_s() - t0;
    ef_phase(EFP_SCHED_START);
    while (more && ln->n_fibers < per_lane) {
      const int sr = start_fiber(w, li);
      if (sr == START_LATER) break;
      more = sr == START_OK;
    }
    ef_phase(EFP_SCHED);
    if (ln->n_fibers > 0) {
      /* run every runnable fibre of the lane until it blocks or ends (the batches of the other
       * lanes are on the GPU meanwhile) */
      t0 = now_s();
      for (size_t i = 0; i < ln->n_fibers; ++i) {
        fiber* f = ln->fibers[i];
        /* a thousand other fibres ran since this one stopped: what it resumes on has left the
         * caches.  While fibre i runs, the top of the stack of the one after the next (the frames of
         * the DP call it returns through) and its answers are fetched. */
        if (prefetch_on && i + 2 < ln->n_fibers) fiber_prefetch(ln->fibers[i + 2]);
        if (f->state == F_RUNNABLE) {
          if (EF_TSAN) tsan_to(f->tsan);
          if (ef_prof_on) { ef_phase(EFP_SCHED); ef_prof.cur = &f->phase; }
          ctx_switch(&w->sched, &f->ctx);
          if (ef_prof_on) { ef_phase(f->phase); ef_prof.cur = NULL; ef_prof.dummy = EFP_SCHED; }
        }
      }
      size_t keep = 0;
      for (size_t i = 0; i < ln->n_fibers; ++i) {
        fiber* f = ln->fibers[i];
        if (f->state == F_DONE) {
          if (memcmp(f->stack, FIBER_SENTINEL, sizeof FIBER_SENTINEL) != 0) {
            fprintf(stderr, "* FATAL fibre stack overflow (raise PINTRON_FIBER_STACK_KB, now %zu)\n", sh->stack_size / 1024);
            abort();
          }
          f->pool_next = w->free_fibers; w->free_fibers = f; w->stats.units++;
        }
        else ln->fibers[keep++] = f;
      }
      ln->n_fibers = keep;
      w->stats.host_s += now_s() - t0;
      t0 = now_s();
      int brc = submit_pairings(w, ln);
      w->stats.pairing_s += now_s() - t0;
      t0 = now_s();
      ef_phase(EFP_SCHED_LAUNCH);
      if (brc == 0) brc = launch_dp(w, ln);
      ef_phase(EFP_SCHED);
      w->stats.dp_s += now_s() - t0;
      if (brc != 0) { sh->failed = 1; break; }
    }
  }
  for (int li = 0; li < n_lanes; ++li) {
    lane* ln = &w->lanes[li];
    collect_dp(w, ln);                   /* only after a failure: do not leave a posted request behind */
    ef_jobbuf_free(&ln->jb);
    free(ln->fibers); free(ln->inflight);
  }
  if (w->ctx) pgpu_destroy(w->ctx);
  if (w->free_fibers && w->index < EF_MAX_WORKERS) { sh->worker_pool[w->index] = w->free_fibers; w->free_fibers = NULL; }
  if (w->free_fibers) {
    fiber* last = w->free_fibers;
    while (last->pool_next) last = last->pool_next;
    pthread_mutex_lock(&sh->mu);
    last->pool_next = sh->fiber_pool; sh->fiber_pool = w->free_fibers;
    pthread_mutex_unlock(&sh->mu);
    w->free_fibers = NULL;
  }
  ef_cell_release_all();
  if (ef_prof_on) {
    ef_phase(EFP_SCHED);
    pthread_mutex_lock(&sh->mu);
    for (int k = 0; k < EFP_N; ++k) { sh->prof_cyc[k] += ef_prof.cyc[k]; sh->prof_susp[k] += ef_prof.susp[k]; sh->prof_jobs[k] += ef_prof.jobs[k]; }
    for (int k = 0; k < 64; ++k) sh->prof_sleep_bins[k] += w->sleep_bins[k];
    sh->prof_ahead[0] += ef_prof.ahead_asked; sh->prof_ahead[1] += ef_prof.ahead_hits; sh->prof_ahead[2] += ef_prof.ahead_misses;
    sh->prof_ahead[3] += ef_prof.chain_graphs; sh->prof_ahead[4] += ef_prof.other_graphs;
    pthread_mutex_unlock(&sh->mu);
  }
  return NULL;
}

static bool env_flag(const char* name) {          /* set and not "0" */
  const char* v = getenv(name);
  return v && !(v[0] == '0' && v[1] == '\0');
}

static size_t env_size(const char* name, size_t dflt) {
  const char* v = getenv(name);
  return (v && atol(v) > 0) ? (size_t)atol(v) : dflt;
}

/* the host cores this process may count on: the online (and allowed) CPUs, cut down to the
 * container's CPU quota when there is one, shared evenly between the ranks torchrun started on
 * this node, and at most the 16 cores one GPU of an 8-GPU node comes with */
static size_t host_core_share(void) {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
  long long quota = -1, period = 0;
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");                       /* cgroup v2: "<quota|max> <period>" */
  if (f) {
    char q[32];
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) != NULL) {     /* cgroup v1 */
    if (fscanf(f, "%lld", &quota) != 1) quota = -1;
    fclose(f);
    if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) != NULL) {
      if (fscanf(f, "%lld", &period) != 1) period = 0;
      fclose(f);
    }
  }
  if (quota > 0 && period > 0) {
    const long q = (long)((quota + period - 1) / period);
    if (q < n) n = q;
  }
  const char* lws = getenv("LOCAL_WORLD_SIZE");
  if (lws && atol(lws) > 1) n /= atol(lws);
  if (n > 16) n = 16;
  return n > 0 ? (size_t)n : 1;
}

/* Worker threads: one per core of the share and an eighth more.  The step is bound by the share's CPU time (C3: 1.40
 * core-seconds per 0.104 s step under a quota of 16 cores, `tools/throttle_check.py`), and a worker is off the CPU a
 * fifth of its time (its batches, the first range of pairings): 18 workers keep 16 cores busy where 16 kept 13.4.
 * Measured on two boxes, alternating (`profiles/r04_sweep_threads_{a,b}.txt`): C3 16 workers / 4 services 104.4 -
 * 107.9 ms, 18 / 6 98.3 - 101.3, 20 / 6 97.1 - 99.2 but with the quota's throttle in 12 of 12 periods (one bad period
 * stalls every thread of the process until the next), 24: 97 - 106; a C5 share 103.6 - 105.5 -> 95.5 - 100.9. */
static size_t default_workers(void) {
  const size_t c = host_core_share();
  return c + c / 8;
}

/* All threads of the step on the socket the GPU hangs off.  On the two-socket hosts of the GPU boxes
 * the unbound program ran 4-14 % slower (threads and their memory spread over both sockets; the
 * per-EST logic is bound by memory latency): the calling thread's affinity is cut down to the CPUs
 * of the GPU's NUMA node, and every thread created from here on inherits it -- the HIP runtime's
 * helpers included, which is why a first guess is made from sysfs before the runtime is up (the
 * dev-th render node under /dev/dri) and corrected by the library's answer afterwards.
 * PINTRON_NUMA=0 leaves the affinity alone, PINTRON_NUMA_NODE=<n> names the node. */
static cpu_set_t numa_original;            /* the affinity the process came with */
static int numa_bound = -2;                /* node the calling thread is bound to; -2: untouched */
static int open_sessions;                  /* the caller's affinity is given back when the last one closes */

static int guess_gpu_numa_node(int dev) {
  /* the dev-th render node is the dev-th HIP device only when no visibility mask renumbers them */
  if (getenv("HIP_VISIBLE_DEVICES") || getenv("ROCR_VISIBLE_DEVICES") || getenv("CUDA_VISIBLE_DEVICES")) return -1;
  int minors[64], n = 0;
  DIR* d = opendir("/dev/dri");
  if (!d) return -1;
  for (struct dirent* e; (e = readdir(d)) != NULL && n < 64;)
    if (strncmp(e->d_name, "renderD", 7) == 0) minors[n++] = atoi(e->d_name + 7);
  closedir(d);
  for (int i = 1; i < n; ++i) for (int j = i; j > 0 && minors[j - 1] > minors[j]; --j) { const int t = minors[j]; minors[j] = minors[j - 1]; minors[j - 1] = t; }
  if (dev < 0 || dev >= n) return -1;
  char path[96];
  snprintf(path, sizeof path, "/sys/class/drm/renderD%d/device/numa_node", minors[dev]);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

/* undo bind_to_numa_node on the calling thread (a library user -- bench.py, pintron_amd.multi -- goes
 * on living after the session: its later threads and child processes must not stay pinned) */
static void restore_affinity(void) {
  if (numa_bound >= -1) sched_setaffinity(0, sizeof numa_original, &numa_original);
  numa_bound = -2;
}

static void bind_to_numa_node(int node) {
  const char* sw = getenv("PINTRON_NUMA");
  if (sw && sw[0] == '0' && sw[1] == '\0') return;
  const char* forced = getenv("PINTRON_NUMA_NODE");
  if (forced && forced[0]) node = atoi(forced);
  if (node < 0 || node == numa_bound) return;
  if (numa_bound == -2 && sched_getaffinity(0, sizeof numa_original, &numa_original) != 0) return;
  char path[96];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return;
  cpu_set_t want;
  CPU_ZERO(&want);
  const char* lws = getenv("LOCAL_WORLD_SIZE");
  /* PINTRON_SMT=1: every hardware thread of the node, as when several ranks share it (tools/sweep_threads.sh) */
  const bool several_ranks = (lws && atoi(lws) > 1) || env_flag("PINTRON_SMT");
  int lo, hi, n = 0;                         /* "a-b,c,d-e" */
  for (;;) {
    if (fscanf(f, "%d", &lo) != 1) break;
    hi = lo;
    int ch = fgetc(f);
    if (ch == '-') { if (fscanf(f, "%d", &hi) != 1) break; ch = fgetc(f); }
    for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c) {
      if (!CPU_ISSET(c, &numa_original)) continue;
      /* one hardware thread per core (the first of its siblings): the workers are compute-bound
       * and hide latency with lanes, two of them on one core only share its pipelines.  Not when
       * several ranks share the node: their threads need the siblings too. */
      if (several_ranks) { CPU_SET(c, &want); ++n; continue; }
      char sp[112];
      snprintf(sp, sizeof sp, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
      FILE* sf = fopen(sp, "r");
      int first = c;
      if (sf) { if (fscanf(sf, "%d", &first) != 1) first = c; fclose(sf); }
      if (first != c && CPU_ISSET(first, &numa_original)) continue;
      CPU_SET(c, &want); ++n;
    }
    if (ch != ',') break;
  }
  fclose(f);
  if (numa_bound == -2) numa_bound = -1;
  if (n >= 2 && sched_setaffinity(0, sizeof want, &want) == 0) numa_bound = node;
}

/* Fibres per worker when PINTRON_FIBERS does not say: enough to cover a batch's round trip with the work of the other
 * fibres, as few as that allows (the fibres' footprint is what makes resumed host code slow).  The work between two
 * suspensions grows with the sequences' length: 768 for ESTs of several hundred bases (C3: 768 -> 114 ms, 1 024 -> 118,
 * 512 -> 119), 1 024 for short reads (a C5 share: 768 -> 111 ms, 1 024 -> 104.5, 1 536 -> 110). */
static size_t default_fibers(const ef_inputs* in) {
  size_t total = 0;
  const size_t n = in->n < 4096 ? in->n : 4096;          /* a sample of the batch */
  for (size_t k = 0; k < n; ++k) total += strlen(in->list[k]->seq);
  return (n && total / n < 300) ? 1024 : 768;
}

/* ---- what outlives a session --------------------------------------------------------------------------
 * A process that runs one batch after the other (est-fact --genes, a library user that brings fresh batches:
 * bench.py's fresh_batch leg) used to take everything apart at the end of a session and make it again for the
 * next: five GPU contexts with their streams, device pools and page-locked buffers, twelve thousand guard-paged
 * fibre stacks, the output chunks, the page-locked slabs of the prefetch stage -- 0.21 s to close and 0.03 s +
 * a slow first step to open, beside a 0.10 s step.  None of it depends on the gene: it is kept here between
 * sessions (PINTRON_KEEP=0: not), handed to the next session that asks, and ends with the process.  (So do the unit
 * table, here, and the slabs of the record arena, in ef_io.c.) */
static struct {
  pthread_mutex_t mu;
  fiber* fibers; size_t stack_size;
  out_chunk* chunks;
  pgpu_ctx* ctx[2 * MAX_SERVICES + 2]; int ctx_device[2 * MAX_SERVICES + 2]; int n_ctx;
  unsigned char* pre_slab; size_t pre_slab_cap; char* up_stage; size_t up_stage_cap;
  void* units; size_t units_cap;            /* the unit table (25 MB for 200 000 entries: 6 000 page faults to get) */
} kept = { PTHREAD_MUTEX_INITIALIZER, NULL, 0, NULL, { NULL }, { 0 }, 0, NULL, 0, NULL, 0, NULL, 0 };
static bool keep_on(void) { const char* e = getenv("PINTRON_KEEP"); return !(e && e[0] == '0' && e[1] == '\0'); }

/* a GPU context on `device`: one kept from an earlier session, else a new one */
static int ctx_take(int device, pgpu_ctx** out) {
  pthread_mutex_lock(&kept.mu);
  for (int k = 0; k < kept.n_ctx; ++k)
    if (kept.ctx_device[k] == device) {
      *out = kept.ctx[k];
      kept.ctx[k] = kept.ctx[kept.n_ctx - 1]; kept.ctx_device[k] = kept.ctx_device[kept.n_ctx - 1]; --kept.n_ctx;
      pthread_mutex_unlock(&kept.mu);
      return PGPU_OK;
    }
  pthread_mutex_unlock(&kept.mu);
  return pgpu_init(device, out);
}
static void ctx_give(pgpu_ctx* ctx, int device) {
  if (!ctx) return;
  if (keep_on()) {
    pthread_mutex_lock(&kept.mu);
    if (kept.n_ctx < (int)(sizeof kept.ctx / sizeof kept.ctx[0])) {
      kept.ctx[kept.n_ctx] = ctx; kept.ctx_device[kept.n_ctx] = device; ++kept.n_ctx;
      pthread_mutex_unlock(&kept.mu);
      return;
    }
    pthread_mutex_unlock(&kept.mu);
  }
  pgpu_destroy(ctx);
}

/* ---- sessions: inputs + index + patterns resident; a step = the whole per-EST pipeline ------------ */
struct ef_session {
  ef_inputs in;
  pgpu_ctx* ctx0;
  shared sh;
  pgpu_pairing_plan* pplan[PRE_CHUNKS];   /* all prepared sequences (both strands), resident in HBM */
  pthread_t pre_thread;
  double pre_kernel_ms[6], pre_meg_ms, pre_t0, pre_wall;
  size_t nthreads;
  double load_s, index_s;
  /* the fibres of the first step (structure + guarded stack) are made while the GPU runtime comes up */
  int device;                              /* the GPU of this session (its contexts go back to `kept` under it) */
  pthread_t pool_thread[2]; int n_pool_threads;
  struct pool_job { shared* sh; size_t count; } pool_job[2];
};

/* Mapping a stack and protecting its guard page are two system calls that take the address-space
 * lock: sixteen workers making a thousand fibres each at the start of the first step queue up
 * behind each other (measured: 0.2 s single-threaded for 16 384 stacks, 0.4 s of WALL with eight
 * threads contending).  Two threads make them here, beside the runtime start-up, touch the pages
 * a fibre starts on and leave them in the shared pool the workers draw from. */
static void* pool_builder_main(void* arg) {
  pthread_setname_np(pthread_self(), "ef-fibre-pool");
  struct pool_job* job = (struct pool_job*)arg;
  shared* sh = job->sh;
  for (size_t i = 0; i < job->count && !sh->failed; ++i) {
    fiber* f = (fiber*)calloc(1, sizeof(fiber));
    if (!f) break;
    f->stack = stack_alloc(sh->stack_size, &f->guarded);
    if (!f->stack) { free(f); break; }
    memcpy(f->stack, FIBER_SENTINEL, sizeof FIBER_SENTINEL);
    ((volatile char*)f->stack)[sh->stack_size - 64] = 0;
    ((volatile char*)f->stack)[sh->stack_size - 4096 - 64] = 0;
    pthread_mutex_lock(&sh->mu);
    f->pool_next = sh->fiber_pool; sh->fiber_pool = f;
    pthread_mutex_unlock(&sh->mu);
  }
  return NULL;
}

/* bringing up the HIP runtime takes a few tenths of a second and the index a tenth: both run
 * beside the parsing and preparation of the ESTs (the genomic sequence is loaded first) */
typedef struct {
  pgpu_ctx* ctx; int rc; const char* gen; size_t gen_len; pgpu_index* idx; int idx_rc;
  int n_svc; pgpu_ctx* svc[MAX_SERVICES]; int svc_rc;     /* the contexts of the service threads come up here too */
  double t_init, t_index, t_svc;                          /* PINTRON_VERBOSE: runtime start-up + first context, index, service contexts */
} gpu_boot;
static void boot_service_contexts(gpu_boot* b) {
  b->svc_rc = PGPU_OK;
  for (int k = 0; k < b->n_svc && b->svc_rc == PGPU_OK; ++k) {
    b->svc_rc = ctx_take(ef_gpu_device_from_env(), &b->svc[k]);
    if (b->svc_rc == PGPU_OK) pgpu_set_timing(b->svc[k], env_flag("PINTRON_KERNEL_TIMING") ? 1 : 0);
  }
}
static void* boot_service_contexts_main(void* arg) {
  gpu_boot* b = (gpu_boot*)arg;
  const double t0 = now_s();
  boot_service_contexts(b);
  b->t_svc = now_s() - t0;
  return NULL;
}
static void* genomic_tables_main(void* arg) {
  pthread_setname_np(pthread_self(), "ef-gene-tables");
  ef_prepare_genomic_tables((ef_inputs*)arg);
  return NULL;
}

static void* gpu_boot_main(void* arg) {
  pthread_setname_np(pthread_self(), "ef-gpu-boot");
  gpu_boot* b = (gpu_boot*)arg;
  const double tb0 = now_s();
  b->rc = ctx_take(ef_gpu_device_from_env(), &b->ctx);
  b->t_init = now_s() - tb0;
  if (b->rc == PGPU_OK) {
    pgpu_set_timing(b->ctx, env_flag("PINTRON_KERNEL_TIMING") ? 1 : 0);
    /* the runtime is up: the service threads' contexts (a stream each) are made beside the index */
    pthread_t svc_thread;
    const bool svc_started = pthread_create(&svc_thread, NULL, boot_service_contexts_main, b) == 0;
    /* PINTRON_INDEX_CACHE=<directory>: the index of a sequence is kept there under its hash and
     * loaded instead of built the next time the same genomic sequence comes by */
    const char* cache = getenv("PINTRON_INDEX_CACHE");
    char path[1200];
    bool loaded = false;
    const double tb1 = now_s();
    if (cache && cache[0]) {
      unsigned long long h = 1469598103934665603ull;
      for (size_t i = 0; i < b->gen_len; ++i) { h ^= (unsigned char)b->gen[i]; h *= 1099511628211ull; }
      snprintf(path, sizeof path, "%s/pintron-index-%016llx-%zu.bin", cache, h, b->gen_len);
      b->idx_rc = pgpu_index_load(b->ctx, path, b->gen, b->gen_len, &b->idx);
      loaded = b->idx_rc == PGPU_OK;
    }
    ef_info_mark("gst-construction-begin");
    if (!loaded) {
      b->idx_rc = pgpu_index_build(b->ctx, b->gen, b->gen_len, &b->idx);
      /* of a sharded run only rank 0 saves (all ranks built the same index) */
      if (b->idx_rc == PGPU_OK && cache && cache[0] && ef_shard_rank == 0 && pgpu_index_save(b->ctx, b->idx, b->gen, path) != PGPU_OK)
        fprintf(stderr, "* WARN the index could not be saved to %s\n", path);
    }
    b->t_index = now_s() - tb1;
    ef_info_mark("gst-preprocessing-begin");            /* (the tables over the suffix array are part of the build here) */
    if (svc_started) pthread_join(svc_thread, NULL); else boot_service_contexts_main(b);
    ef_info_mark("gst-preprocessing-end");
  }
  return NULL;
}

ef_session* ef_session_open(int argc, char** argv) {
  const double t_start = now_s();
  ef_info_mark("start");
  /* the per-EST code allocates and frees a few hundred small blocks per EST on every worker; keep
   * the arenas from returning memory to the system and asking for it again between ESTs */
  mallopt(M_TRIM_THRESHOLD, 512 << 20);
  mallopt(M_TOP_PAD, 16 << 20);
  bind_to_numa_node(guess_gpu_numa_node(ef_gpu_device_from_env()));   /* before the runtime starts its threads */
  ef_session* s = (ef_session*)calloc(1, sizeof(ef_session));
  ++open_sessions;
  pthread_mutex_init(&s->sh.mu, NULL);
  pthread_cond_init(&s->sh.ready_cv, NULL);
  pthread_mutex_init(&s->sh.svc.mu, NULL);
  pthread_cond_init(&s->sh.svc.posted, NULL);
  pthread_cond_init(&s->sh.svc.finished, NULL);
  int load_rc = ef_load_genomic_sequence(argc, argv, &s->in);
  if (load_rc != 0) { ef_session_close(s); return NULL; }
  gpu_boot boot;
  memset(&boot, 0, sizeof boot);
  boot.rc = boot.idx_rc = boot.svc_rc = PGPU_EDEVICE;
  boot.gen = s->in.gen->seq; boot.gen_len = strlen(s->in.gen->seq);
  /* four service threads, no coalescing wait, eight lanes per worker: since a batch became one launch on one
   * stream (round 3) many small batches beat few large ones -- +9 % on C3, +14 % on a C5 share against
   * 3 / 50 us / 4 (profiles/r03_sweep_sched_*.txt) */
  {                                          /* ... three service threads per eight cores of this rank's share, at most six
                                              * (round 4, with the workers' count: see default_workers) */
    size_t dflt = host_core_share() * 3 / 8;
    if (dflt < 1) dflt = 1;
    if (dflt > 6) dflt = 6;
    boot.n_svc = (int)env_size("PINTRON_SERVICES", dflt);
  }
  if (boot.n_svc > MAX_SERVICES) boot.n_svc = MAX_SERVICES;
  pthread_t boot_thread;
  const bool booting = pthread_create(&boot_thread, NULL, gpu_boot_main, &boot) == 0;
  ef_parse_threads = (int)env_size("PINTRON_PARSE_THREADS", host_core_share());      /* the cores idle while the GPU runtime starts */
  /* the per-gene tables (6-mer index, splice-site score and class tables: 0.025 s for 200 kb) are made beside the
   * reading and preparation of the ESTs (0.025 s for 100 000): neither looks at the other's data */
  {
    pthread_t tab_thread;
    const bool tab_started = pthread_create(&tab_thread, NULL, genomic_tables_main, &s->in) == 0;
    if (!tab_started) ef_prepare_genomic_tables(&s->in);
    load_rc = ef_load_ests(&s->in);
    if (tab_started) pthread_join(tab_thread, NULL);
  }
  ef_classify_init();
  const double t_loaded = now_s();
  ef_info_mark("data-io-end");
  s->device = ef_gpu_device_from_env();
  size_t kept_fibers = 0;
  {                                            /* what an earlier session of this process left (see `kept`) */
    const size_t stack_size = env_size("PINTRON_FIBER_STACK_KB", 256) * 1024;
    pthread_mutex_lock(&kept.mu);
    if (kept.fibers && kept.stack_size == stack_size) { s->sh.fiber_pool = kept.fibers; kept.fibers = NULL; }
    s->sh.spare_chunks = kept.chunks; kept.chunks = NULL;
    s->sh.pre_slab = kept.pre_slab; s->sh.pre_slab_cap = kept.pre_slab_cap; kept.pre_slab = NULL; kept.pre_slab_cap = 0;
    s->sh.up_stage = kept.up_stage; s->sh.up_stage_cap = kept.up_stage_cap; kept.up_stage = NULL; kept.up_stage_cap = 0;
    pthread_mutex_unlock(&kept.mu);
    for (fiber* f = s->sh.fiber_pool; f; f = f->pool_next) ++kept_fibers;
  }
  if (load_rc == 0 && !getenv("PINTRON_NO_FIBER_POOL")) {
    s->sh.stack_size = env_size("PINTRON_FIBER_STACK_KB", 256) * 1024;
    size_t want = env_size("PINTRON_THREADS", default_workers()) * env_size("PINTRON_FIBERS", default_fibers(&s->in));
    if (want > s->in.n) want = s->in.n;                 /* never more fibres than sequences */
    want = want > kept_fibers ? want - kept_fibers : 0;
    if (want >= 64) {
      for (int t = 0; t < 2; ++t) {
        s->pool_job[t].sh = &s->sh; s->pool_job[t].count = want / 2;
        if (pthread_create(&s->pool_thread[s->n_pool_threads], NULL, pool_builder_main, &s->pool_job[t]) == 0) ++s->n_pool_threads;
      }
    }
  }
  if (booting) pthread_join(boot_thread, NULL); else gpu_boot_main(&boot);
  const double t_booted = now_s();
  if (boot.rc == PGPU_OK) { s->ctx0 = boot.ctx; s->sh.idx = boot.idx_rc == PGPU_OK ? boot.idx : NULL; }
  for (int k = 0; k < boot.n_svc; ++k) s->sh.svc.threads[k].ctx = boot.svc[k];     /* (closed with the session) */
  if (load_rc != 0) { ef_session_close(s); return NULL; }
  if (boot.rc != PGPU_OK) {
    fprintf(stderr, "* FATAL no usable MI355X (gfx950) device / libpintron_gpu.so: est-fact has no CPU fallback\n");
    ef_session_close(s); return NULL;
  }
  bind_to_numa_node(pgpu_device_numa_node(s->ctx0));   /* the library's word on the first guess */
  ef_inputs* in = &s->in;
  shared* sh = &s->sh;
  sh->in = in;
  if (boot.idx_rc != PGPU_OK) {
    fprintf(stderr, "* FATAL pgpu_index_build: %s\n", pgpu_last_error(s->ctx0));
    ef_session_close(s); return NULL;
  }
  {
    pthread_mutex_lock(&kept.mu);
    if (kept.units && kept.units_cap >= in->n + 1) { sh->units = (unit*)kept.units; sh->units_cap = kept.units_cap; kept.units = NULL; kept.units_cap = 0; }
    pthread_mutex_unlock(&kept.mu);
    if (sh->units) memset(sh->units, 0, (in->n + 1) * sizeof(unit));
    else { sh->units = (unit*)calloc(in->n + 1, sizeof(unit)); sh->units_cap = in->n + 1; }
  }
  for (size_t k = 0; k < in->n;) {
    unit* u = &sh->units[sh->n_units++];
    u->first = k;
    u->has_sibling = in->has_rev ? in->has_rev[k] != 0 : !in->list[k]->fixed_strand;
    k += u->has_sibling ? 2 : 1;
  }
  if (!getenv("PINTRON_NO_PREFETCH") && in->n > 0) {
    /* ranges of whole units with about the same number of entries (the workers wait for the first one only;
     * small first ranges -- 1/32, 1/32, 1/16, then eighths -- were tried and lost 2.5 %: profiles/r03_sweep_chunks.txt) */
    /* ... and a range is a dozen kernel launches and two waits: a small batch is cut into fewer (a thousand ESTs in ten
     * ranges spent 24 of their 28 ms there) */
    size_t want = env_size("PINTRON_PRE_CHUNKS", 10);
    if (!getenv("PINTRON_PRE_CHUNKS")) { const size_t by_size = sh->n_units / 4096 + 1; if (by_size < want) want = by_size; }
    if (want > PRE_CHUNKS) want = PRE_CHUNKS;
    sh->n_pre = sh->n_units < want ? (int)sh->n_units : (int)want;
    /* PINTRON_PRE_RAMP="w0,w1,...": relative sizes of the ranges (their number then follows from the list), for
     * measurements.  Round 4, after a worker stopped sleeping on a range that is not back while it has fibres to run
     * (start_fiber): first ranges of 1 - 3 % of the batch still lose (C3 105.7 / 107.1 ms, 104.3 / 103.7, 101.3 / 102.8
     * against 102.8 / 99.7 with ten equal ranges, alternating on one box: the first batches of the step are then a few
     * jobs each); a first and last range of half the size measured 97.5 / 96.6 on that box and 96.4 / 102.7 / 98.9
     * against 97.4 / 97.1 / 101.2 on the next, a C5 share 104.3 / 109.4 against 98.7 / 99.7: equal ranges stay
     * (`profiles/r04_sweep_ramp_{a,b}.txt`). */
    double wts[PRE_CHUNKS]; int nw = 0;
    const char* ramp = getenv("PINTRON_PRE_RAMP");
    if (ramp && ramp[0]) {
      for (const char* q = ramp; *q && nw < PRE_CHUNKS;) {
        char* end = NULL;
        const double v = strtod(q, &end);
        if (end == q) break;
        if (v > 0) wts[nw++] = v;
        if (*end != ',' && *end != ':') break;
        q = end + 1;
      }
    }
    if (nw > 0 && (size_t)nw <= sh->n_units) sh->n_pre = nw;
    else { nw = sh->n_pre; for (int k = 0; k < nw; ++k) wts[k] = 1.0; }
    double tot = 0, acc = 0;
    for (int k = 0; k < nw; ++k) tot += wts[k];
    for (int cidx = 0; cidx <= sh->n_pre; ++cidx) {
      size_t u = cidx == sh->n_pre ? sh->n_units : (size_t)((double)sh->n_units * (acc / tot));
      if (cidx > 0 && cidx < sh->n_pre) {                      /* never an empty range */
        size_t prev_u = 0;
        while (prev_u < sh->n_units && sh->units[prev_u].first < sh->pre_lo[cidx - 1]) ++prev_u;
        if (u <= prev_u) u = prev_u + 1;
      }
      if (u > sh->n_units) u = sh->n_units;
      sh->pre_lo[cidx] = u < sh->n_units ? sh->units[u].first : in->n;
      if (cidx < sh->n_pre) acc += wts[cidx];
    }
  }
  sh->svc.n_threads = boot.n_svc;
  sh->svc.coalesce_us = getenv("PINTRON_COALESCE_US") ? atol(getenv("PINTRON_COALESCE_US")) : 0;
  if (boot.svc_rc != PGPU_OK) {
    fprintf(stderr, "* FATAL the GPU contexts of the service threads could not be created\n");
    ef_session_close(s); return NULL;
  }
  if (getenv("PINTRON_VERBOSE"))
    fprintf(stderr, "* open: load %.3fs, then GPU runtime + index + service contexts still %.3fs, rest %.3fs (GPU side: runtime + first context %.3fs, index %.3fs, service contexts %.3fs)\n",
            t_loaded - t_start, t_booted - t_loaded, now_s() - t_booted, boot.t_init, boot.t_index, boot.t_svc);
  /* the workers hide the GPU latency with lanes, not with oversubscription */
  s->nthreads = env_size("PINTRON_THREADS", default_workers());
  if (s->nthreads > sh->n_units) s->nthreads = sh->n_units ? sh->n_units : 1;
  sh->max_fibers = env_size("PINTRON_FIBERS", default_fibers(in));
  sh->stack_size = env_size("PINTRON_FIBER_STACK_KB", 256) * 1024;
  sh->kernel_timing = env_flag("PINTRON_KERNEL_TIMING");
  sh->gen_len = strlen(in->gen->seq);
  {
    const char* gm = getenv("PINTRON_GPU_MEG");            /* 0: the graphs are built on the host from the pairings */
    sh->use_meg = !(gm && gm[0] == '0' && gm[1] == '\0');
    const ef_config* c = &in->cfg;
    const pgpu_meg_params mp = { c->min_factor_len, c->min_intron_length, c->max_intron_length, c->max_pairings_in_MEG,
                                 c->max_prefix_discarded_rate, c->max_suffix_discarded_rate, c->max_freq_shortest_pairing,
                                 c->trans_red ? 1u : 0u, c->short_edge_comp ? 1u : 0u };
    sh->meg_prm = mp;
  }
  sh->n_lanes = (int)env_size("PINTRON_LANES", 8);
  if (sh->n_lanes > MAX_LANES) sh->n_lanes = MAX_LANES;
  s->load_s = t_loaded - t_start;
  s->index_s = now_s() - t_loaded;
  return s;
}

/* the workers' own fibre lists back into the shared one (before the session is taken apart or measured) */
static void merge_worker_pools(shared* sh) {
  for (size_t t = 0; t < EF_MAX_WORKERS; ++t) {
    fiber* f = sh->worker_pool[t];
    sh->worker_pool[t] = NULL;
    while (f) { fiber* nx = f->pool_next; f->pool_next = sh->fiber_pool; sh->fiber_pool = f; f = nx; }
  }
}

/* forget the previous step's output; its chunks are kept for the next step unless `release` */
static void free_unit_buffers(shared* sh, bool release) {
  for (size_t u = 0; u < sh->n_units; ++u)
    for (int k = 0; k < EF_N_OUT; ++k) { sh->units[u].buf[k] = NULL; sh->units[u].len[k] = 0; }
  while (sh->chunks) { out_chunk* nx = sh->chunks->next; sh->chunks->next = sh->spare_chunks; sh->spare_chunks = sh->chunks; sh->chunks = nx; }
  if (release) while (sh->spare_chunks) { out_chunk* nx = sh->spare_chunks->next; free(sh->spare_chunks); sh->spare_chunks = nx; }
}

/* one pass of the whole hot path over the batch: pairing prefetch (resident patterns), then the
 * fibres (MEG, embeddings, DP batches, refinement) on all worker threads */
/* prefetch thread: the pairings of chunk after chunk (one resident batch each); every finished
 * chunk releases its units to the workers */
/* the prepared sequences (both strands) of chunk c, concatenated, as a resident pairing plan */
/* The host half of a range's plan (offsets + the sequences copied into the staging buffer) is made by a helper thread
 * while the prefetch thread has the range before it on the device (first step only): a fresh batch's prefetch stage was
 * 0.105 s against 0.056 s with the patterns resident, and 2.5 ms per range of it were these two loops. */
typedef struct { ef_session* s; int c; uint64_t* off; size_t total; char* blob; bool own_blob; int rc; pthread_t th; bool started; } stage_job;
static void* stage_main(void* arg) {
  stage_job* j = (stage_job*)arg;
  ef_session* s = j->s;
  shared* sh = &s->sh;
  ef_inputs* in = &s->in;
  const size_t lo = sh->pre_lo[j->c], hi = sh->pre_lo[j->c + 1];
  j->rc = PGPU_OK; j->blob = NULL; j->own_blob = false;
  j->off = (uint64_t*)malloc((hi - lo + 1) * sizeof(uint64_t));
  if (!j->off) { j->rc = PGPU_ENOMEM; return NULL; }
  size_t total = 0;
  for (size_t k = lo; k < hi; ++k) { j->off[k - lo] = total; total += strlen(in->list[k]->seq); }
  j->off[hi - lo] = total;
  j->total = total;
  if (total + 1 > sh->up_stage_cap) {                       /* page-locked: the copy to the device runs at PCIe speed */
    if (sh->up_stage) pgpu_host_free(s->ctx0, sh->up_stage);
    sh->up_stage = NULL; sh->up_stage_cap = 0;
    void* q = NULL;
    const size_t want = total + total / 4 + 4096;
    if (pgpu_host_alloc(s->ctx0, want, &q) == PGPU_OK) { sh->up_stage = (char*)q; sh->up_stage_cap = want; }
  }
  j->blob = sh->up_stage ? sh->up_stage : (char*)malloc(total + 1);
  j->own_blob = j->blob != sh->up_stage;
  if (!j->blob) { j->rc = PGPU_ENOMEM; return NULL; }
  for (size_t k = lo; k < hi; ++k) memcpy(j->blob + j->off[k - lo], in->list[k]->seq, (size_t)(j->off[k - lo + 1] - j->off[k - lo]));
  return NULL;
}
/* start staging range c beside the caller (the staging buffer must be free: the plan of the range before has been made) */
static void stage_start(stage_job* j, ef_session* s, int c) {
  memset(j, 0, sizeof *j);
  j->s = s; j->c = c;
  j->started = pthread_create(&j->th, NULL, stage_main, j) == 0;
  if (!j->started) stage_main(j);
}
/* the device half: the staged sequences as a resident pairing plan */
static int stage_finish(stage_job* j) {
  if (j->started) { pthread_join(j->th, NULL); j->started = false; }
  ef_session* s = j->s;
  int prc = j->rc;
  if (prc == PGPU_OK) {
    const size_t lo = s->sh.pre_lo[j->c], hi = s->sh.pre_lo[j->c + 1];
    prc = pgpu_pairing_plan_create_resident(s->ctx0, s->sh.idx, j->blob, j->off, hi - lo, &s->pplan[j->c]);   /* returns after the copy */
  }
  if (j->own_blob) free(j->blob);
  free(j->off);
  j->off = NULL; j->blob = NULL;
  return prc;
}

static void* prefetch_main(void* arg) {
  pthread_setname_np(pthread_self(), "ef-pairings");
  ef_session* s = (ef_session*)arg;
  shared* sh = &s->sh;
  pgpu_pairing_params prm = { s->in.cfg.min_factor_len, 0, s->in.cfg.min_string_depth_rate };
  stage_job stage; bool staging = false;
  memset(&stage, 0, sizeof stage);
  for (int c = 0; c < sh->n_pre; ++c) {
    char rname[48];
    snprintf(rname, sizeof rname, "prefetch chunk %d (pairings + MEGs)", c);
    pgpu_range_push(rname);
    /* first step: the chunk's sequences go to the device here, chunk after chunk beside the workers
     * that already factorize the chunks before (they stay resident for the steps that follow) */
    const double tc0 = now_s();
    int prc = PGPU_OK;
    if (!s->pplan[c]) {
      if (!staging) { stage_start(&stage, s, c); staging = true; }       /* (the first range: nothing to hide behind) */
      prc = stage_finish(&stage);
      staging = false;
      /* the next range's sequences are copied together while this one is on the device */
      if (prc == PGPU_OK && c + 1 < sh->n_pre && !s->pplan[c + 1]) { stage_start(&stage, s, c + 1); staging = true; }
    }
    const double tc1 = now_s();
    if (prc == PGPU_OK) prc = pgpu_pairing_plan_run(s->ctx0, s->pplan[c], &prm);
    const double tc2 = now_s();
    bool have_meg = false;
    if (prc == PGPU_OK && sh->use_meg) {
      /* the graphs are built where the pairings lie; only the finished records cross PCIe */
      const int mrc = pgpu_pairing_plan_run_meg(s->ctx0, s->pplan[c], &sh->meg_prm);
      if (mrc == PGPU_ENOSYS) sh->use_meg = false;           /* a library without the MEG stage */
      else if (mrc != PGPU_OK) prc = mrc;
      else {
        const size_t bytes = (size_t)pgpu_pairing_plan_meg_bytes(s->pplan[c]);
        const size_t entries = sh->pre_lo[c + 1] - sh->pre_lo[c];
        if (!sh->pre_slab && c == 1 && sh->n_pre > 2) {
          /* the first chunk got blocks of its own -- the workers wait for it -- and the rest are about as large as
           * the second: their memory is allocated now, in one piece, while the workers have the first to do */
          const size_t per = (bytes + bytes / 4 + 4096 + (entries + 2) * sizeof(uint64_t) + 511) & ~(size_t)511;
          void* q = NULL;
          if (pgpu_host_alloc(s->ctx0, per * (size_t)(sh->n_pre - 1), &q) == PGPU_OK) { sh->pre_slab = (unsigned char*)q; sh->pre_slab_cap = per * (size_t)(sh->n_pre - 1); }
        }
        if (bytes > sh->pre_meg_cap[c] || !sh->pre_meg[c]) {
          if (sh->pre_meg[c] && sh->pre_own[c]) pgpu_host_free(s->ctx0, sh->pre_meg[c]);
          sh->pre_meg[c] = NULL; sh->pre_own[c] = false;
          sh->pre_meg_cap[c] = bytes + bytes / 8 + 4096;
          const size_t need = (sh->pre_meg_cap[c] + 255) & ~(size_t)255;
          if (sh->pre_slab && sh->pre_slab_used + need <= sh->pre_slab_cap) { sh->pre_meg[c] = sh->pre_slab + sh->pre_slab_used; sh->pre_slab_used += need; }
          else {
            void* q = NULL;
            if (pgpu_host_alloc(s->ctx0, sh->pre_meg_cap[c], &q) == PGPU_OK) { sh->pre_meg[c] = (unsigned char*)q; sh->pre_own[c] = true; }
          }
        }
        if (!sh->pre_meg_first[c]) {
          const size_t need = ((entries + 1) * sizeof(uint64_t) + 255) & ~(size_t)255;
          if (sh->pre_slab && sh->pre_slab_used + need <= sh->pre_slab_cap) { sh->pre_meg_first[c] = (uint64_t*)(sh->pre_slab + sh->pre_slab_used); sh->pre_slab_used += need; }
          else {
            void* q = NULL;
            if (pgpu_host_alloc(s->ctx0, (entries + 1) * sizeof(uint64_t), &q) == PGPU_OK) { sh->pre_meg_first[c] = (uint64_t*)q; sh->pre_first_own[c] = true; }
          }
        }
        if (!sh->pre_meg[c] || !sh->pre_meg_first[c]) prc = PGPU_ENOMEM;
        else prc = pgpu_pairing_plan_fetch_meg(s->ctx0, s->pplan[c], sh->pre_meg[c], sh->pre_meg_cap[c], sh->pre_meg_first[c]);
        have_meg = prc == PGPU_OK;
        s->pre_meg_ms += pgpu_pairing_plan_meg_ms(s->pplan[c]);
      }
    }
    if (prc == PGPU_OK && !have_meg) {
      const size_t cnt = (size_t)pgpu_pairing_plan_count(s->pplan[c]);
      sh->pre_tri[c] = (pgpu_pairing*)malloc((cnt + 1) * sizeof(pgpu_pairing));
      sh->pre_first[c] = (uint64_t*)malloc((sh->pre_lo[c + 1] - sh->pre_lo[c] + 1) * sizeof(uint64_t));
      prc = pgpu_pairing_plan_fetch(s->ctx0, s->pplan[c], sh->pre_tri[c], cnt, sh->pre_first[c]);
    }
    for (int k = 0; k < 6; ++k) s->pre_kernel_ms[k] += pgpu_pairing_plan_kernel_ms(s->pplan[c], k);
    if (getenv("PINTRON_VERBOSE") && atoi(getenv("PINTRON_VERBOSE")) >= 2)
      fprintf(stderr, "* prefetch chunk %d: at %.3fs of the step; plan + upload %.1f ms, pairings %.1f ms, MEGs + download %.1f ms\n", c,
              tc0 - s->pre_t0, 1e3 * (tc1 - tc0), 1e3 * (tc2 - tc1), 1e3 * (now_s() - tc2));
    pthread_mutex_lock(&sh->mu);
    if (prc != PGPU_OK) { fprintf(stderr, "* FATAL pairing prefetch failed: %s\n", pgpu_last_error(s->ctx0)); sh->failed = 1; }
    else __atomic_store_n(&sh->ready_entries, sh->pre_lo[c + 1], __ATOMIC_RELEASE);
    pthread_cond_broadcast(&sh->ready_cv);
    pthread_mutex_unlock(&sh->mu);
    pgpu_range_pop();
    if (prc != PGPU_OK) break;
  }
  if (staging) {                               /* left behind by a failure: nobody will use it */
    if (stage.started) pthread_join(stage.th, NULL);
    if (stage.own_blob) free(stage.blob);
    free(stage.off);
  }
  s->pre_wall = now_s() - s->pre_t0;
  return NULL;
}

typedef struct { double wall, user, sys; long minflt; } run_mark;
static run_mark run_mark_now(void) {
  struct rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  run_mark m = { now_s(), ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec, ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec, ru.ru_minflt };
  return m;
}

static int cmp_interval(const void* x, const void* y) {
  const double a = *(const double*)x, b = *(const double*)y;
  return a < b ? -1 : a > b ? 1 : 0;
}

int ef_session_step(ef_session* s, ef_sched_stats* stats_out) {
  shared* sh = &s->sh;
  const double t0 = now_s();
  pgpu_range_push("est-fact step");
  const bool step_rusage = getenv("PINTRON_STEP_RUSAGE") != NULL;
  run_mark ru0; if (step_rusage) ru0 = run_mark_now();
  free_unit_buffers(sh, false);
  for (int c = 0; c < PRE_CHUNKS; ++c) { free(sh->pre_tri[c]); free(sh->pre_first[c]); sh->pre_tri[c] = NULL; sh->pre_first[c] = NULL; }
  sh->next_unit = 0; sh->failed = 0; sh->ready_entries = 0;
  sh->prof_t0 = t0; memset(sh->prof_sleep_bins, 0, sizeof sh->prof_sleep_bins); memset(sh->prof_ahead, 0, sizeof sh->prof_ahead);
  memset(s->pre_kernel_ms, 0, sizeof s->pre_kernel_ms);
  s->pre_meg_ms = 0;
  s->pre_t0 = t0; s->pre_wall = 0;
  bool pre_started = false;
  if (sh->n_pre) {
    pre_started = pthread_create(&s->pre_thread, NULL, prefetch_main, s) == 0;
    if (!pre_started) prefetch_main(s);                       /* no thread to be had: in line */
  }
  const double t1 = now_s();
  service* sv = &sh->svc;
  sv->stop = false; sv->head = sv->tail = NULL; sv->sh = sh;
  for (int k = 0; k < sv->n_threads; ++k) {
    memset(&sv->threads[k].stats, 0, sizeof(ef_sched_stats));
    memset(sv->threads[k].phase_s, 0, sizeof sv->threads[k].phase_s);
    sv->threads[k].n_iv = 0;
    sv->threads[k].sv = sv;
  }
  int sv_started = 0;
  while (sv_started < sv->n_threads && pthread_create(&sv->threads[sv_started].thread, NULL, service_main, &sv->threads[sv_started]) == 0) ++sv_started;
  worker* ws = (worker*)calloc(s->nthreads, sizeof(worker));
  pthread_t* th = (pthread_t*)malloc(s->nthreads * sizeof(pthread_t));
  size_t w_started = 0;
  if (sv_started > 0)            /* a worker that cannot get a thread is simply not started: the others take its ESTs */
    for (size_t t = 0; t < s->nthreads; ++t) { ws[w_started].sh = sh; ws[w_started].index = w_started; if (pthread_create(&th[w_started], NULL, worker_main, &ws[w_started]) == 0) ++w_started; }
  if (sv_started == 0 || w_started == 0) { fprintf(stderr, "* FATAL cannot start the service / worker threads\n"); sh->failed = 1; }
  const unsigned long long prof_c0 = ef_prof_now(); const double prof_t0 = now_s();
  memset(sh->prof_cyc, 0, sizeof sh->prof_cyc); memset(sh->prof_susp, 0, sizeof sh->prof_susp); memset(sh->prof_jobs, 0, sizeof sh->prof_jobs);
  for (size_t t = 0; t < w_started; ++t) pthread_join(th[t], NULL);
  if (ef_prof_on && w_started) {
    static const char* nm[EFP_N] = { "other", "meg", "embeddings", "endpoints", "external", "dust", "noisy", "add", "filters", "gap-errors",
                                     "refine-intron", "tail/polyA", "refinement", "output", "side-files", "free", "scheduler", "sleep",
                                     "ref:affixes", "ref:false-small", "ref:new-small", "ref:clean", "sched:launch", "sched:collect", "sched:start", "wait:prefetch",
                                     "new-small:prefix", "new-small:ask", "new-small:class", "new-small:search", "search:lookups", "search:decide", "search:end" };
    const double hz = (double)(ef_prof_now() - prof_c0) / (now_s() - prof_t0);
    unsigned long long ts = 0, tj = 0; double tot = 0;
    fprintf(stderr, "* profile (per worker thread, %zu workers, %zu units): phase, seconds, suspensions / unit, jobs / unit\n", w_started, sh->n_units);
    for (int k = 0; k < EFP_N; ++k) {
      const double sec = (double)sh->prof_cyc[k] / hz / (double)w_started;
      fprintf(stderr, "*   %-14s %8.4f s  %6.2f  %6.2f\n", nm[k], sec, (double)sh->prof_susp[k] / (double)sh->n_units, (double)sh->prof_jobs[k] / (double)sh->n_units);
      ts += sh->prof_susp[k]; tj += sh->prof_jobs[k]; if (k != EFP_SLEEP && k != EFP_WAIT_PREFETCH) tot += sec;
    }
    fprintf(stderr, "*   asked ahead: %.2f jobs / unit, found there %.2f, asked in turn (while answers were kept) %.2f\n",
            (double)sh->prof_ahead[0] / (double)sh->n_units, (double)sh->prof_ahead[1] / (double)sh->n_units, (double)sh->prof_ahead[2] / (double)sh->n_units);
    fprintf(stderr, "*   graphs that are one path (enumerated from the record): %llu of %llu\n", sh->prof_ahead[3], sh->prof_ahead[3] + sh->prof_ahead[4]);
    fprintf(stderr, "*   asleep waiting for batches, ms per worker in each 5 ms of the step:");
    for (int k = 0; k < 64 && k * 0.005 < now_s() - t0; ++k) fprintf(stderr, " %.1f", 1e3 * sh->prof_sleep_bins[k] / (double)w_started);
    fprintf(stderr, "\n");
    fprintf(stderr, "*   %-14s %8.4f s  %6.2f  %6.2f\n", "all but sleeps", tot, (double)ts / (double)sh->n_units, (double)tj / (double)sh->n_units);
  }
  if (pre_started) pthread_join(s->pre_thread, NULL);
  pthread_mutex_lock(&sv->mu); sv->stop = true; pthread_cond_broadcast(&sv->posted); pthread_mutex_unlock(&sv->mu);
  for (int k = 0; k < sv_started; ++k) pthread_join(sv->threads[k].thread, NULL);
  ef_sched_stats st;
  memset(&st, 0, sizeof st);
  st.threads = s->nthreads;
  for (int t = 0; t < sv_started; ++t) {
    const ef_sched_stats* ss = &sv->threads[t].stats;
    const double* ph = sv->threads[t].phase_s;
    if (getenv("PINTRON_VERBOSE"))
      fprintf(stderr, "* service %d: %zu batches, %zu jobs; idle %.3fs merge %.3fs create %.3fs launch %.3fs sync %.3fs fetch %.3fs\n",
              t, ss->dp_batches, ss->dp_jobs, ph[0], ph[1] - ph[0], ph[2], ph[3], ph[4], ph[5]);
    st.dp_batches += ss->dp_batches; st.dp_jobs += ss->dp_jobs;
    for (int k = 0; k < ss->n_kernels; ++k) kstat_add(&st, &ss->kernels[k]);
  }
  {                                       /* the launches of all service threads on one time line */
    size_t n = 0;
    for (int t = 0; t < sv_started; ++t) n += sv->threads[t].n_iv / 2;
    if (n) {
      double* a = (double*)malloc(2 * n * sizeof(double));
      size_t k = 0;
      for (int t = 0; t < sv_started; ++t) { memcpy(a + k, sv->threads[t].iv, sv->threads[t].n_iv * sizeof(double)); k += sv->threads[t].n_iv; }
      qsort(a, n, 2 * sizeof(double), cmp_interval);
      double busy = 0, lo = a[0], hi = a[1];
      for (size_t i = 1; i < n; ++i) {
        if (a[2 * i] > hi) { busy += hi - lo; lo = a[2 * i]; hi = a[2 * i + 1]; }
        else if (a[2 * i + 1] > hi) hi = a[2 * i + 1];
      }
      st.dp_busy_union_ms = busy + (hi - lo);
      free(a);
    }
  }
  for (size_t t = 0; t < w_started; ++t) {
    st.units += ws[t].stats.units;
    st.pairing_batches += ws[t].stats.pairing_batches; st.pairing_requests += ws[t].stats.pairing_requests;
    st.host_s += ws[t].stats.host_s; st.pairing_s += ws[t].stats.pairing_s; st.dp_s += ws[t].stats.dp_s;
    for (int k = 0; k < ws[t].stats.n_kernels; ++k) kstat_add(&st, &ws[t].stats.kernels[k]);
  }
  if (sh->n_pre) {
    static const char* nm[6] = { "pair_locate", "pair_chain", "pair_count+scan", "pair_fill", "pair_cross+scan", "pair_emit" };
    for (int k = 0; k < 6; ++k) {
      ef_kernel_stat ks; memset(&ks, 0, sizeof ks);
      snprintf(ks.name, sizeof ks.name, "%s", nm[k]);
      ks.ms = s->pre_kernel_ms[k]; ks.launches = (size_t)sh->n_pre; ks.jobs = s->in.n;
      /* algorithmic bytes of the table-guided search (restated for the 8-mer
       * table): per position its pattern byte, one 2 x 4 B probe of the 4^8-entry table, two
       * bisections over the run of that 8-mer (|T| / 4^8 suffixes on average: ceil(log2(run + 1))
       * steps of one 4 B suffix-array probe each) and 3 x 4 B written (interval, longest match); plus
       * one 4 B suffix-array read per occurrence reported.  12 bytes per pairing written by emit. */
      if (k == 0 || k == 5) {
        unsigned long long positions = 0, pairs = 0;
        for (int c = 0; c < sh->n_pre; ++c) { positions += pgpu_pairing_plan_positions(s->pplan[c]); pairs += pgpu_pairing_plan_count(s->pplan[c]); }
        unsigned long long run = sh->gen_len / 65536ull + 1;
        unsigned lg = 0;
        while ((1ull << lg) < run + 1) ++lg;
        ks.algo_bytes = k == 0 ? positions * (1ull + 8ull + 2ull * 4ull * lg + 12ull) + pairs * 4ull : pairs * 12ull;
      }
      if (ks.ms > 0) kstat_add(&st, &ks);
    }
    if (s->pre_meg_ms > 0) {
      ef_kernel_stat ks; memset(&ks, 0, sizeof ks);
      snprintf(ks.name, sizeof ks.name, "meg_build+emit");
      ks.ms = s->pre_meg_ms; ks.launches = (size_t)sh->n_pre; ks.jobs = s->in.n;
      /* algorithmic bytes: the pairings read (12 B each) and the records written */
      unsigned long long pairs = 0, rec = 0;
      for (int c = 0; c < sh->n_pre; ++c) { pairs += pgpu_pairing_plan_count(s->pplan[c]); rec += pgpu_pairing_plan_meg_bytes(s->pplan[c]); }
      ks.algo_bytes = pairs * 12ull + rec;
      kstat_add(&st, &ks);
    }
  }
  { size_t su = 0; for (size_t t = 0; t < w_started; ++t) su += ws[t].suspensions; st.suspensions_per_unit = sh->n_units ? (double)su / (double)sh->n_units : 0.0; }
  st.load_s = s->load_s; st.index_s = s->index_s; st.prefetch_s = s->pre_wall; st.workers_s = now_s() - t1;
  for (size_t u = 0; u < sh->n_units; ++u) if (sh->units[u].len[1]) ++st.aligned;
  if (stats_out) *stats_out = st;
  ef_dp_trace_flush();
  pgpu_range_pop();
  ef_info_mark("est-processing-end");
  free(ws); free(th);
  if (step_rusage) {
    const run_mark ru1 = run_mark_now();
    fprintf(stderr, "* step: %.3fs wall, user %.2fs sys %.2fs (all threads of the process), %ld page faults\n",
            ru1.wall - ru0.wall, ru1.user - ru0.user, ru1.sys - ru0.sys, ru1.minflt - ru0.minflt);
  }
  return sh->failed ? 1 : 0;
}

/* the six files of the step, in input order, into the current directory: every file is cut into ranges of
 * units, a range is one task (gathering pwritev calls at the offset the lengths before it give), and a few
 * threads take the tasks -- the largest file alone (raw-multifasta-out, ~150 MB on C3) no longer sets the time.
 * (Filling the files through shared mappings from all threads was tried and is slower: C5 0.85 - 1.0 s against
 * 0.69 - 0.92 s, C3 0.08 against 0.05 s -- tools/exp/oneshot_ab.sh.) */
enum { WRITE_RANGES = 8, WRITE_THREADS_MAX = 16, WRITE_IOV = 512 };
typedef struct { int fd, k; size_t u0, u1; off_t at; } write_task;
typedef struct { shared* sh; write_task* tasks; int n_tasks; int next; int failed; } write_pool;
static void* write_pool_main(void* arg) {
  pthread_setname_np(pthread_self(), "ef-writer");
  write_pool* wp = (write_pool*)arg;
  struct iovec iov[WRITE_IOV];
  for (;;) {
    const int t = __atomic_fetch_add(&wp->next, 1, __ATOMIC_RELAXED);
    if (t >= wp->n_tasks) return NULL;
    const write_task* wt = &wp->tasks[t];
    off_t at = wt->at;
    size_t u = wt->u0;
    while (u < wt->u1) {
      int n = 0; size_t bytes = 0;
      for (; u < wt->u1 && n < WRITE_IOV; ++u) {
        const unit* un = &wp->sh->units[u];
        if (!un->len[wt->k]) continue;
        iov[n].iov_base = un->buf[wt->k]; iov[n].iov_len = un->len[wt->k]; bytes += un->len[wt->k]; ++n;
      }
      int first = 0;
      while (bytes) {                                   /* a short write continues where it stopped */
        const ssize_t w = pwritev(wt->fd, iov + first, n - first, at);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) { __atomic_store_n(&wp->failed, 1, __ATOMIC_RELAXED); return NULL; }   /* 0: quota / odd filesystem, never spin on it */
        at += w; bytes -= (size_t)w;
        size_t left = (size_t)w;
        while (left && first < n) {
          if (left >= iov[first].iov_len) { left -= iov[first].iov_len; ++first; }
          else { iov[first].iov_base = (char*)iov[first].iov_base + left; iov[first].iov_len -= left; left = 0; }
        }
      }
    }
  }
}

/* units' text of file k (0..5, 6 = records) through a stream, in input order */
typedef struct { shared* sh; FILE* f; int k; } file_writer;
static void* file_writer_main(void* arg) {
  file_writer* fw = (file_writer*)arg;
  setvbuf(fw->f, NULL, _IOFBF, 1 << 20);
  for (size_t u = 0; u < fw->sh->n_units; ++u)
    if (fw->sh->units[u].len[fw->k]) fwrite(fw->sh->units[u].buf[fw->k], 1, fw->sh->units[u].len[fw->k], fw->f);
  return NULL;
}

int ef_session_write_outputs(ef_session* s) {
  ef_outputs out;
  if (ef_open_outputs(&out)) return 1;
  shared* sh = &s->sh;
  FILE* dst[6] = { out.fout.f, out.fests.f, out.fmeg.f, out.fpmeg.f, out.ftmeg.f, out.fintronic.f };
  write_task tasks[6 * WRITE_RANGES];
  write_pool wp = { sh, tasks, 0, 0, 0 };
  for (int k = 0; k < 6; ++k) {
    fflush(dst[k]);
    off_t at = 0;
    for (int r = 0; r < WRITE_RANGES; ++r) {
      const size_t u0 = sh->n_units * (size_t)r / WRITE_RANGES, u1 = sh->n_units * (size_t)(r + 1) / WRITE_RANGES;
      size_t bytes = 0;
      for (size_t u = u0; u < u1; ++u) bytes += sh->units[u].len[k];
      if (bytes) { write_task wt = { fileno(dst[k]), k, u0, u1, at }; tasks[wp.n_tasks++] = wt; }
      at += (off_t)bytes;
    }
  }
  /* largest tasks first: the ranges of one file are about equal, the files are not */
  {
    size_t w[6] = {0};
    for (int k = 0; k < 6; ++k) for (size_t u = 0; u < sh->n_units; u += 64) w[k] += sh->units[u].len[k];
    for (int a = 1; a < wp.n_tasks; ++a) {
      const write_task t = tasks[a]; int b = a;
      while (b > 0 && w[tasks[b - 1].k] < w[t.k]) { tasks[b] = tasks[b - 1]; --b; }
      tasks[b] = t;
    }
  }
  {
    size_t nth = env_size("PINTRON_WRITE_THREADS", host_core_share());
    if (nth > WRITE_THREADS_MAX) nth = WRITE_THREADS_MAX;
    if (nth > (size_t)wp.n_tasks) nth = (size_t)wp.n_tasks;
    pthread_t th[WRITE_THREADS_MAX]; size_t started = 0;
    for (size_t t = 1; t < nth; ++t) if (pthread_create(&th[started], NULL, write_pool_main, &wp) == 0) ++started;
    write_pool_main(&wp);
    for (size_t t = 0; t < started; ++t) pthread_join(th[t], NULL);
  }
  /* PINTRON_RECORDS_FILE=<path>: the packed factorization records (include/pintron_records.h) too */
  const char* rec_path = getenv("PINTRON_RECORDS_FILE");
  int rec_rc = 0;
  if (rec_path && rec_path[0]) {
    FILE* rf = fopen(rec_path, "wb");
    if (!rf) { fprintf(stderr, "* cannot write %s\n", rec_path); rec_rc = 1; }
    else {
      file_writer rw = { sh, rf, 6 };
      file_writer_main(&rw);
      if (fclose(rf) != 0) rec_rc = 1;
    }
  }
  ef_close_outputs(&out);
  if (wp.failed) { fprintf(stderr, "* FATAL writing the output files failed\n"); return 1; }
  return rec_rc;
}

/* text of output file `which` of the last step (0 raw-multifasta-out, 1 processed-ests, 2 megs,
 * 3 processed-megs, 4 processed-megs-info, 5 meg-edges; 6 = the packed factorization records of
 * ef_write_factorization_records), concatenated in input order; caller frees */
char* ef_session_output(ef_session* s, int which, size_t* len) {
  shared* sh = &s->sh;
  if (which < 0 || which >= EF_N_OUT) { *len = 0; return NULL; }
  size_t total = 0;
  for (size_t u = 0; u < sh->n_units; ++u) total += sh->units[u].len[which];
  char* r = (char*)malloc(total + 1);
  size_t pos = 0;
  for (size_t u = 0; u < sh->n_units; ++u) { memcpy(r + pos, sh->units[u].buf[which], sh->units[u].len[which]); pos += sh->units[u].len[which]; }
  r[pos] = '\0';
  *len = total;
  return r;
}

/* raw-multifasta-out records of the last step */
char* ef_session_records(ef_session* s, size_t* len) { return ef_session_output(s, 0, len); }

size_t ef_session_n_ests(const ef_session* s) { return s->sh.n_units; }
struct pgpu_ctx* ef_session_context(ef_session* s) { return s->ctx0; }
const void* ef_session_genomic(ef_session* s) { return s->in.gen; }

void ef_session_close(ef_session* s) {
  if (!s) return;
  shared* sh = &s->sh;
  const bool verbose = getenv("PINTRON_VERBOSE") != NULL;
  double tq[6]; tq[0] = now_s();
  for (int t = 0; t < s->n_pool_threads; ++t) pthread_join(s->pool_thread[t], NULL);
  s->n_pool_threads = 0;
  free_unit_buffers(sh, !keep_on());
  if (keep_on() && sh->spare_chunks) {
    pthread_mutex_lock(&kept.mu);
    out_chunk* last = sh->spare_chunks;
    while (last->next) last = last->next;
    last->next = kept.chunks; kept.chunks = sh->spare_chunks; sh->spare_chunks = NULL;
    pthread_mutex_unlock(&kept.mu);
  }
  merge_worker_pools(sh);
  if (getenv("PINTRON_STACK_STATS")) {      /* how deep did the fibres' stacks get? (first byte written above the sentinel) */
    size_t n = 0, sum = 0, mx = 0;
    for (fiber* f = sh->fiber_pool; f; f = f->pool_next) {
      size_t lo = sizeof FIBER_SENTINEL;
      while (lo < sh->stack_size && f->stack[lo] == 0) lo += 64;
      const size_t used = lo < sh->stack_size ? sh->stack_size - lo : 0;
      ++n; sum += used; if (used > mx) mx = used;
    }
    fprintf(stderr, "* fibre stacks: %zu fibres, %zu B used on average, %zu B at most (of %zu)\n", n, n ? sum / n : 0, mx, sh->stack_size);
  }
  if (keep_on() && sh->fiber_pool && sh->stack_size) {           /* the fibres (stacks, sink blocks) wait for the next session */
    pthread_mutex_lock(&kept.mu);
    if (!kept.fibers || kept.stack_size == sh->stack_size) {
      fiber* last = sh->fiber_pool;
      while (last->pool_next) last = last->pool_next;
      last->pool_next = kept.fibers; kept.fibers = sh->fiber_pool; kept.stack_size = sh->stack_size;
      sh->fiber_pool = NULL;
    }
    pthread_mutex_unlock(&kept.mu);
  }
  while (sh->fiber_pool) {
    fiber* nx = sh->fiber_pool->pool_next;
    for (int k = 0; k < EF_N_OUT; ++k) free(sh->fiber_pool->out[k].mem);
    stack_free(sh->fiber_pool->stack, sh->stack_size, sh->fiber_pool->guarded); free(sh->fiber_pool);
    sh->fiber_pool = nx;
  }
  if (keep_on() && sh->units) {
    pthread_mutex_lock(&kept.mu);
    if (!kept.units || kept.units_cap < sh->units_cap) { free(kept.units); kept.units = sh->units; kept.units_cap = sh->units_cap; sh->units = NULL; }
    pthread_mutex_unlock(&kept.mu);
  }
  free(sh->units);
  tq[1] = now_s();
  for (int k = 0; k < MAX_SERVICES; ++k) { free(sh->svc.threads[k].iv); sh->svc.threads[k].iv = NULL; sh->svc.threads[k].n_iv = sh->svc.threads[k].cap_iv = 0; }
  for (int c = 0; c < PRE_CHUNKS && s->ctx0; ++c) {
    free(sh->pre_tri[c]); free(sh->pre_first[c]);
    if (sh->pre_meg[c] && sh->pre_own[c]) pgpu_host_free(s->ctx0, sh->pre_meg[c]);
    if (sh->pre_meg_first[c] && sh->pre_first_own[c]) pgpu_host_free(s->ctx0, sh->pre_meg_first[c]);
    if (s->pplan[c]) pgpu_pairing_plan_destroy(s->ctx0, s->pplan[c]);
  }
  if (keep_on() && s->ctx0) {                  /* the page-locked slabs too (0.25 s per GB to get) */
    pthread_mutex_lock(&kept.mu);
    if (sh->pre_slab && !kept.pre_slab) { kept.pre_slab = sh->pre_slab; kept.pre_slab_cap = sh->pre_slab_cap; sh->pre_slab = NULL; }
    if (sh->up_stage && !kept.up_stage) { kept.up_stage = sh->up_stage; kept.up_stage_cap = sh->up_stage_cap; sh->up_stage = NULL; }
    pthread_mutex_unlock(&kept.mu);
  }
  tq[2] = now_s();
  if (sh->pre_slab) pgpu_host_free(s->ctx0, sh->pre_slab);
  if (sh->up_stage) pgpu_host_free(s->ctx0, sh->up_stage);
  if (s->ctx0 && sh->idx) pgpu_index_destroy(s->ctx0, sh->idx);
  tq[3] = now_s();
  for (int k = 0; k < MAX_SERVICES; ++k) ctx_give(sh->svc.threads[k].ctx, s->device);
  ctx_give(s->ctx0, s->device);
  tq[4] = now_s();
  ef_free_inputs(&s->in);
  tq[5] = now_s();
  if (verbose)
    fprintf(stderr, "* close: unit buffers + fibres %.3fs, pairing plans %.3fs, index + slabs %.3fs, contexts %.3fs, inputs %.3fs\n",
            tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4]);
  pthread_mutex_destroy(&sh->mu); pthread_cond_destroy(&sh->ready_cv);
  pthread_mutex_destroy(&sh->svc.mu); pthread_cond_destroy(&sh->svc.posted); pthread_cond_destroy(&sh->svc.finished);
  free(s);
  if (--open_sessions == 0) restore_affinity();
}

int ef_leave_without_cleanup = 0;

/* A process that ends hands its pages back to the kernel one after the other, on one core, at ~0.1 s per GB:
 * 0.3 s of a one-shot C3 run, 0.8 s of a full C5 input.  What is this program's own and large -- the record
 * arena, the text of the output files, the fibre stacks, the unit table -- is dropped from all cores first
 * (madvise(MADV_DONTNEED) needs the address-space lock only for reading: 0.08 s for 4 GB, tools/exp/exit_cost).
 * Nothing of it is read afterwards: the files are written, the statistics copied.  The memory of malloc's
 * arenas and of the GPU runtime is left alone. */
typedef struct { char* p; size_t len; } page_range;
typedef struct { page_range* r; size_t n; size_t next; } page_release;
static void* page_release_main(void* arg) {
  page_release* pr = (page_release*)arg;
  for (;;) {
    const size_t k = __atomic_fetch_add(&pr->next, 1, __ATOMIC_RELAXED);
    if (k >= pr->n) return NULL;
    madvise(pr->r[k].p, pr->r[k].len, MADV_DONTNEED);
  }
}
static void session_release_pages(ef_session* s) {
  shared* sh = &s->sh;
  merge_worker_pools(sh);
  size_t cap = 1024, n = 0;
  for (fiber* f = sh->fiber_pool; f; f = f->pool_next) ++cap;
  for (out_chunk* c = sh->chunks; c; c = c->next) ++cap;
  for (out_chunk* c = sh->spare_chunks; c; c = c->next) ++cap;
  void** ab = (void**)malloc(65536 * sizeof(void*)); size_t* al = (size_t*)malloc(65536 * sizeof(size_t));
  if (!ab || !al) { free(ab); free(al); return; }
  const size_t na = ef_record_arena_regions(s->in.arena, ab, al, 65536);
  page_range* r = (page_range*)malloc((cap + na) * 16 * sizeof(page_range));
  if (!r) { free(ab); free(al); return; }
  const size_t piece = (size_t)8 << 20, room = (cap + na) * 16;
#define ADD_RANGE(ptr, bytes) do { \
    uintptr_t a_ = ((uintptr_t)(ptr) + 4095) & ~(uintptr_t)4095, b_ = ((uintptr_t)(ptr) + (bytes)) & ~(uintptr_t)4095; \
    for (; a_ < b_ && n < room; a_ += piece) { r[n].p = (char*)a_; r[n].len = b_ - a_ < piece ? b_ - a_ : piece; ++n; } } while (0)
  for (size_t k = 0; k < na; ++k) ADD_RANGE(ab[k], al[k]);
  for (out_chunk* c = sh->chunks; c; c = c->next) ADD_RANGE(c->data, c->cap);
  for (out_chunk* c = sh->spare_chunks; c; c = c->next) ADD_RANGE(c->data, c->cap);
  for (fiber* f = sh->fiber_pool; f; f = f->pool_next) if (f->guarded) ADD_RANGE(f->stack, sh->stack_size);
  if (sh->units) ADD_RANGE(sh->units, (sh->n_units + 1) * sizeof(unit));
#undef ADD_RANGE
  page_release pr = { r, n, 0 };
  size_t nth = host_core_share();
  if (nth > 16) nth = 16;
  pthread_t th[16]; size_t started = 0;
  for (size_t t = 1; t < nth; ++t) if (pthread_create(&th[started], NULL, page_release_main, &pr) == 0) ++started;
  page_release_main(&pr);
  for (size_t t = 0; t < started; ++t) pthread_join(th[t], NULL);
  free(r); free(ab); free(al);
}

int ef_run_batched_stats(int argc, char** argv, ef_sched_stats* stats_out) {
  const run_mark m0 = run_mark_now();
  ef_session* s = ef_session_open(argc, argv);
  if (!s) return 1;
  ef_sched_stats st;
  const run_mark m1 = run_mark_now();
  int rc = ef_session_step(s, &st);
  const run_mark m2 = run_mark_now();
  if (rc == 0) rc = ef_session_write_outputs(s);
  const run_mark m3 = run_mark_now();
  if (stats_out) *stats_out = st;
  /* The files are on disk and every stream has been waited for.  A process that is about to
   * exit (ef_leave_without_cleanup, set by the est-fact program) does not take the session apart
   * (200 000 sequences, fibre stacks, device pools: 0.3 s): the caller ends it with _exit. */
  const bool leave = ef_leave_without_cleanup != 0;
  if (!leave) ef_session_close(s);
  else if (!getenv("PINTRON_NO_PAGE_RELEASE")) session_release_pages(s);
  ef_log_reference_timers(st.index_s, st.prefetch_s, st.workers_s, st.load_s + (m3.wall - m2.wall), now_s() - m0.wall);
  if (getenv("PINTRON_VERBOSE")) {
    fprintf(stderr, "* run: open %.3fs step %.3fs write %.3fs close %.3fs\n", m1.wall - m0.wall, m2.wall - m1.wall, m3.wall - m2.wall, now_s() - m3.wall);
    fprintf(stderr, "* cpu (all threads): open user %.2fs sys %.2fs, %ld page faults; step user %.2fs sys %.2fs, %ld page faults; write user %.2fs sys %.2fs\n",
            m1.user - m0.user, m1.sys - m0.sys, m1.minflt - m0.minflt, m2.user - m1.user, m2.sys - m1.sys, m2.minflt - m1.minflt,
            m3.user - m2.user, m3.sys - m2.sys);
  }
  return rc;
}

int ef_run_batched(int argc, char** argv) {
  ef_sched_stats st;
  const int rc = ef_run_batched_stats(argc, argv, &st);
  if (rc == 0 && getenv("PINTRON_VERBOSE")) {
    fprintf(stderr, "est-fact: %zu ESTs (%zu aligned), %zu threads, %zu pairing batches (%zu requests), %zu DP batches (%zu jobs); "
                    "load %.2fs index %.2fs prefetch %.2fs workers %.2fs [per-thread avg: host %.2fs pairing %.2fs dp %.2fs]\n",
            st.units, st.aligned, st.threads, st.pairing_batches, st.pairing_requests, st.dp_batches, st.dp_jobs,
            st.load_s, st.index_s, st.prefetch_s, st.workers_s, st.host_s / st.threads, st.pairing_s / st.threads, st.dp_s / st.threads);
    for (int k = 0; k < st.n_kernels; ++k)
      fprintf(stderr, "  kernel %-28s launches %6zu jobs %9zu  %9.3f ms  %8.1f algo-GB/s\n", st.kernels[k].name, st.kernels[k].launches,
              st.kernels[k].jobs, st.kernels[k].ms, st.kernels[k].ms > 0 ? st.kernels[k].algo_bytes / (st.kernels[k].ms * 1e-3) / 1e9 : 0.0);
  }
  return rc;
}
