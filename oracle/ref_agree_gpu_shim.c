/*
 * Reference-side binding shown in INTEGRATION.md section 8: the three dynamic
 * programs the reference's intron-agreement stage calls -- compute_alignment (src/agree-introns.c:
 * 629,695), edit_distance (:762, src/main-intron-agreement.c:857,864; only the last cell of the
 * matrix is read at every one of these sites) and compute_gap_alignment (:837) -- answered by
 * libpintron_gpu.so through its C-ABI, one call at a time (the minimal,
 * synchronous binding).  The same ALIGN / ED / GAP kernels as est-fact, unchanged.
 *
 * Built by oracle/Makefile into _ref/intron-agreement-gpu: the reference's own intron-agreement
 * sources from where they lie + this file, linked with -Wl,--wrap=<routine> so that every call of
 * the three routines from the reference's code lands here.  tests/test_gpu_intron_agreement.py runs
 * it beside the unmodified intron-agreement-ref: predicted-introns.txt and
 * out-after-intron-agree.txt must be identical.  Not part of the product, never linked into it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "types.h"
#include "list.h"
#include "util.h"

#include "../include/pintron_gpu.h"
#ifdef AGREE_SELF_CHECK      /* debugging build: every device answer is compared with the CPU oracle */
#include "dp_oracle.h"
#endif

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <fcntl.h>
/* PINTRON_AGREE_DEBUG: where a crash of the bound program happens (return addresses + the map of the
 * executable, to be resolved with addr2line on the build machine) */
static void crash_report(int sig) {
  void* bt[48];
  const int n = backtrace(bt, 48);
  backtrace_symbols_fd(bt, n, 2);
  char buf[4096];
  const int fd = open("/proc/self/maps", O_RDONLY);
  if (fd >= 0) { const ssize_t k = read(fd, buf, sizeof buf); if (k > 0) write(2, buf, (size_t)k); close(fd); }
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void debug_setup(void) {
  if (getenv("PINTRON_AGREE_DEBUG")) { signal(SIGSEGV, crash_report); fprintf(stderr, "DBG shim loaded\n"); }
}

static pgpu_ctx* ctx;
static unsigned long n_calls[3];          /* ALIGN, ED, GAP answered by the device */

static void report(void) {
  if (getenv("PINTRON_VERBOSE"))
    fprintf(stderr, "intron-agreement-gpu: %lu alignments, %lu edit distances, %lu gap alignments on the device\n",
            n_calls[0], n_calls[1], n_calls[2]);
}

static void need_ctx(void) {
  if (ctx) return;
  atexit(report);
  const char* d = getenv("PINTRON_GPU_DEVICE");
  if (pgpu_init(d ? atoi(d) : 0, &ctx) != PGPU_OK) {
    fprintf(stderr, "* FATAL intron-agreement-gpu: no usable MI355X (gfx950) device\n");
    exit(1);
  }
}

/* one job, both operands in a private arena; strings (if any) into a buffer the caller frees */
static pgpu_dp_result run_one(uint32_t kind, const char* a, size_t la, const char* b, size_t lb, char** strings) {
  if (getenv("PINTRON_AGREE_DEBUG")) fprintf(stderr, "DBG enter kind %u %zu x %zu\n", kind, la, lb);
  need_ctx();
  char* arena = (char*)malloc(la + lb + 8);
  memcpy(arena, a, la); memcpy(arena + la, b, lb);
  pgpu_dp_job job;
  memset(&job, 0, sizeof job);
  job.kind = kind; job.a_off = 0; job.a_len = (uint32_t)la; job.b_off = la; job.b_len = (uint32_t)lb;
  const size_t cap = 2 * (la + lb + 1) + 64;
  char* str = strings ? (char*)malloc(cap) : NULL;
  pgpu_dp_result res;
  if (pgpu_dp_batch(ctx, NULL, &job, 1, arena, la + lb, &res, str, str ? cap : 0, NULL) != PGPU_OK || res.status != PGPU_OK) {
    fprintf(stderr, "* FATAL intron-agreement-gpu: DP job of kind %u (%zu x %zu) failed: %s\n", kind, la, lb, pgpu_last_error(ctx));
    exit(1);
  }
  free(arena);
#ifdef AGREE_SELF_CHECK
  {
    char* o = (char*)calloc(2 * (la + lb + 2), 1);
    int bad = 0;
    if (kind == PGPU_DP_ALIGN) {
      int32_t dim; const int32_t sc = (int32_t)orc_align(a, la, b, lb, o, o + la + lb + 2, &dim);
      bad = sc != res.v[0] || dim != res.v[1] || memcmp(o, str + res.str[0], (size_t)dim) || memcmp(o + la + lb + 2, str + res.str[1], (size_t)dim);
      if (bad) fprintf(stderr, "SELF-CHECK ALIGN differs: score %d/%d dim %d/%d\n a=%.*s\n b=%.*s\n", sc, res.v[0], dim, res.v[1], (int)la, a, (int)lb, b);
    } else if (kind == PGPU_DP_GAP) {
      orc_gap_result g; orc_gap_align(a, la, b, lb, o, o + la + lb + 2, &g);
      bad = g.dim != res.v[0] || g.factor_cut != res.v[1] || g.intron_start != res.v[2] || g.intron_end != res.v[3] ||
            g.intron_start_on_align != res.v[4] || g.intron_end_on_align != res.v[5] || memcmp(o, str + res.str[0], (size_t)g.dim) || memcmp(o + la + lb + 2, str + res.str[1], (size_t)g.dim);
      if (bad) fprintf(stderr, "SELF-CHECK GAP differs: dim %d/%d cut %d/%d is %d/%d ie %d/%d\n a=%.*s\n b=%.*s\n", g.dim, res.v[0], g.factor_cut, res.v[1], g.intron_start, res.v[2], g.intron_end, res.v[3], (int)la, a, (int)lb, b);
    } else {
      const int32_t d = (int32_t)orc_edit_distance(a, la, b, lb);
      bad = d != res.v[0];
      if (bad) fprintf(stderr, "SELF-CHECK ED differs: %d/%d\n a=%.*s\n b=%.*s\n", d, res.v[0], (int)la, a, (int)lb, b);
    }
    free(o);
  }
#endif
  if (strings) *strings = str;
  return res;
}

plist __wrap_compute_alignment(char* EST_seq, char* genomic_seq, bool only_one_align) {
  fail_if(!only_one_align);
  const size_t n = strlen(EST_seq), m = strlen(genomic_seq);
  char* str = NULL;
  const pgpu_dp_result r = run_one(PGPU_DP_ALIGN, EST_seq, n, genomic_seq, m, &str);
  ++n_calls[0];
  if (getenv("PINTRON_AGREE_DEBUG")) fprintf(stderr, "DBG A %d %d %zu %zu\n", r.v[0], r.v[1], n, m);
  palignment al = alignment_create(n + m + 1);
  al->score = r.v[0];
  al->alignment_dim = r.v[1];
  memcpy(al->EST_alignment, str + r.str[0], (size_t)r.v[1]); al->EST_alignment[r.v[1]] = '\0';
  memcpy(al->GEN_alignment, str + r.str[1], (size_t)r.v[1]); al->GEN_alignment[r.v[1]] = '\0';
  free(str);
  plist out = list_create();
  list_add_to_tail(out, al);
  return out;
}

/* every caller in intron-agreement reads M[(l1+1)*(l2+1)-1] and frees M */
unsigned int* __wrap_edit_distance(char* s1, size_t l1, char* s2, size_t l2) {
  const pgpu_dp_result r = run_one(PGPU_DP_ED, s1, l1, s2, l2, NULL);
  ++n_calls[1];
  if (getenv("PINTRON_AGREE_DEBUG")) fprintf(stderr, "DBG E %d %zu %zu\n", r.v[0], l1, l2);
  unsigned int* M = (unsigned int*)calloc((l1 + 1) * (l2 + 1), sizeof(unsigned int));
  M[(l1 + 1) * (l2 + 1) - 1] = (unsigned int)r.v[0];
  return M;
}

plist __wrap_compute_gap_alignment(char* EST_seq, char* genomic_seq, bool only_one_align, int gen_cut, int gen_cut_left, int gen_cut_right) {
  (void)only_one_align; (void)gen_cut; (void)gen_cut_left; (void)gen_cut_right;     /* src/refine-intron.c:565-569 */
  const size_t n = strlen(EST_seq), m = strlen(genomic_seq);
  char* str = NULL;
  const pgpu_dp_result r = run_one(PGPU_DP_GAP, EST_seq, n, genomic_seq, m, &str);
  ++n_calls[2];
  if (getenv("PINTRON_AGREE_DEBUG")) fprintf(stderr, "DBG G %d %d %d %d %d %d %zu %zu %u %u %.*s %.*s\n", r.v[0], r.v[1], r.v[2], r.v[3], r.v[4], r.v[5], n, m, r.str[0], r.str[1], r.v[0], str + r.str[0], r.v[0], str + r.str[1]);
  pgap_alignment g = gap_alignment_create(n + m + 10);
  g->gap_alignment_dim = r.v[0]; g->factor_cut = r.v[1]; g->intron_start = r.v[2]; g->intron_end = r.v[3];
  g->intron_start_on_align = r.v[4]; g->intron_end_on_align = r.v[5];
  memcpy(g->EST_gap_alignment, str + r.str[0], (size_t)r.v[0]); g->EST_gap_alignment[r.v[0]] = '\0';
  memcpy(g->GEN_gap_alignment, str + r.str[1], (size_t)r.v[0]); g->GEN_gap_alignment[r.v[0]] = '\0';
  free(str);
  plist out = list_create();
  list_add_to_tail(out, g);
  return out;
}
