/* GPU box experiment: host memory (RssAnon = the queues' context-save areas, RssShmem = streams + page-locked
 * buffers) and time per library context, made on one thread or on a thread each. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../../include/pintron_gpu.h"
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void rss(const char* what, double dt) {
  FILE* f = fopen("/proc/self/status", "r"); char line[256]; long anon = 0, shm = 0;
  while (f && fgets(line, sizeof line, f)) { if (!strncmp(line, "RssAnon:", 8)) anon = atol(line + 8); if (!strncmp(line, "RssShmem:", 9)) shm = atol(line + 9); }
  if (f) fclose(f);
  fprintf(stderr, "%-44s %.3f s  RssAnon %5ld MB  RssShmem %4ld MB\n", what, dt, anon / 1024, shm / 1024);
}
static void* one_ctx(void* arg) {
  pgpu_ctx* c = NULL;
  if (pgpu_init(0, &c) != PGPU_OK) { fprintf(stderr, "pgpu_init failed\n"); return NULL; }
  if (arg) {                              /* use it: a tiny index build runs kernels and copies on its stream */
    pgpu_index* ix = NULL;
    const char* g = "ACGTACGTTAGCATCGATCGATTACGATCGATCGGCTAGCTAGCATCGATCGACTAGCTAGCATGCATGCAGTCAGT";
    pgpu_index_build(c, g, strlen(g), &ix);
  }
  return c;
}
int main(int argc, char** argv) {
  const int use = argc > 1 ? atoi(argv[1]) : 1;
  double t = now();
  rss("start", 0);
  one_ctx(use ? (void*)1 : NULL); rss("first context (runtime start-up included)", now() - t);
  for (int k = 0; k < 4; ++k) { t = now(); one_ctx(use ? (void*)1 : NULL); rss("one more context, same thread", now() - t); }
  for (int k = 0; k < 4; ++k) {
    t = now(); pthread_t th; pthread_create(&th, NULL, one_ctx, use ? (void*)1 : NULL); pthread_join(th, NULL);
    rss("one more context, on a thread of its own", now() - t);
  }
  return 0;
}
