/* TEST INFRASTRUCTURE: times the host logic + fibre scheduler of est-fact without a GPU.
 * Runs the session step twice over the inputs of the current directory on the CPU stand-in of the
 * C-ABI (fake_pgpu.c) with its answer cache on: the first pass fills the cache (oracle DP), the
 * following passes find every answer there, so their time is host code only.
 *   PINTRON_FAKE_CACHE=1 PINTRON_THREADS=1 ./sched_profile [passes] */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "../../pintron_amd/host/ef_sched.h"

extern int moncontrol(int);   /* gprof builds: sample the cached passes only */
#pragma weak moncontrol

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double cpu_s(void) { struct timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv) {
  const int passes = argc > 1 ? atoi(argv[1]) : 3;
  setenv("PINTRON_FAKE_CACHE", "1", 1);
  char* av[2] = { (char*)"est-fact", NULL };
  if (moncontrol) moncontrol(0);            /* loading and the per-gene tables are not the steady state */
  ef_session* s = ef_session_open(1, av);
  if (!s) return 1;
  for (int p = 0; p < passes; ++p) {
    ef_sched_stats st;
    if (moncontrol) moncontrol(p > 0);
    const double t0 = now_s(), c0 = cpu_s();
    if (ef_session_step(s, &st) != 0) return 1;
    fprintf(stderr, "pass %d: %.3f s wall, %.3f s cpu, %zu ESTs, %zu DP jobs in %zu batches -> %.1f us cpu per EST\n",
            p, now_s() - t0, cpu_s() - c0, st.units, st.dp_jobs, st.dp_batches, 1e6 * (cpu_s() - c0) / (double)st.units);
  }
  ef_session_close(s);
  return 0;
}
