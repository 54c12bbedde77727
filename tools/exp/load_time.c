/* experiment: the host-side load of est-fact (genomic tables, parse, preparation) alone, with CPU times and faults */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <sys/resource.h>
#include "../../pintron_amd/host/estfact.h"
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void mark(const char* what, double* t, struct rusage* r) {
  struct rusage r1; getrusage(RUSAGE_SELF, &r1);
  const double t1 = now();
  fprintf(stderr, "  %-28s %.3f s wall, user %.2f sys %.2f, %ld faults\n", what, t1 - *t,
          (r1.ru_utime.tv_sec - r->ru_utime.tv_sec) + 1e-6 * (r1.ru_utime.tv_usec - r->ru_utime.tv_usec),
          (r1.ru_stime.tv_sec - r->ru_stime.tv_sec) + 1e-6 * (r1.ru_stime.tv_usec - r->ru_stime.tv_usec), r1.ru_minflt - r->ru_minflt);
  *t = t1; *r = r1;
}
int main(int argc, char** argv) {
  ef_inputs in;
  ef_parse_threads = argc > 1 ? atoi(argv[1]) : 8;
  char* av[] = { "x", NULL };
  struct rusage r; getrusage(RUSAGE_SELF, &r);
  double t = now();
  fprintf(stderr, "threads %d\n", ef_parse_threads);
  if (ef_load_genomic_sequence(1, av, &in)) return 1;
  mark("genomic sequence", &t, &r);
  ef_prepare_genomic_tables(&in);
  mark("genomic tables", &t, &r);
  ef_seq** ests = NULL;
  in.arena = ef_record_arena_new();
  const long n = ef_read_multifasta_arena("ests.txt", 0, 1, getenv("NO_ARENA") ? NULL : in.arena, &ests);
  mark("read + parse", &t, &r);
  free(ests);
  in.arena = NULL;
  if (ef_load_ests(&in)) return 1;
  mark("ef_load_ests (all of it, again)", &t, &r);
  fprintf(stderr, "  %ld records, %zu entries\n", n, in.n);
  return 0;
}
