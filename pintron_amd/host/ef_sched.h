/* Batched (fibre) execution of est-fact over the GPU C-ABI: see ef_sched.c */
#ifndef EF_SCHED_H
#define EF_SCHED_H
#include <stddef.h>

typedef struct {
  size_t threads, units, dp_batches, dp_jobs, pairing_batches, pairing_requests;
  double load_s, index_s, workers_s;      /* wall-clock phases */
  double host_s, pairing_s, dp_s;         /* summed over threads: fibres / pairing batches / DP batches */
} ef_sched_stats;

/* environment: PINTRON_THREADS (default: online CPUs), PINTRON_FIBERS (fibres per thread, 2048),
 * PINTRON_FIBER_STACK_KB (256), PINTRON_GPU_DEVICE (0), PINTRON_VERBOSE */
int ef_run_batched(int argc, char** argv);
int ef_run_batched_stats(int argc, char** argv, ef_sched_stats* stats);

#endif
