/* Batched (fibre) execution of est-fact over the GPU C-ABI: see ef_sched.c */
#ifndef EF_SCHED_H
#define EF_SCHED_H
#include <stddef.h>

#define EF_MAX_KERNELS 64
typedef struct {
  char name[48];
  double ms;                               /* summed HIP-event time of all launches */
  size_t launches, jobs;
  unsigned long long cells, algo_bytes;
} ef_kernel_stat;

typedef struct {
  size_t threads, units, aligned, dp_batches, dp_jobs, pairing_batches, pairing_requests;
  double load_s, index_s, prefetch_s, workers_s;   /* wall-clock phases (prefetch runs beside the workers) */
  double host_s, pairing_s, dp_s;         /* summed over threads: fibres / pairing batches / DP batches */
  int n_kernels;                          /* filled when PINTRON_KERNEL_TIMING is set */
  ef_kernel_stat kernels[EF_MAX_KERNELS];
  /* PINTRON_KERNEL_TIMING: the time the device spent in the DP batches' kernels with the launches of all service
   * threads laid on one time line (union of their [start, end) intervals) -- the kernels' `ms` are sums over
   * launches that overlap in time, this is what can be compared with the step's wall time; 0 = not measured */
  double dp_busy_union_ms;
  double suspensions_per_unit;            /* times an EST gave up its thread to wait for answers, per input EST */
} ef_sched_stats;

/* session = inputs of the current directory loaded, genomic index and all prepared sequences
 * resident in HBM; step = one pass of the whole est-fact hot path over the batch */
typedef struct ef_session ef_session;
ef_session* ef_session_open(int argc, char** argv);
int ef_session_step(ef_session* s, ef_sched_stats* stats);
int ef_session_write_outputs(ef_session* s);
char* ef_session_records(ef_session* s, size_t* len);
char* ef_session_output(ef_session* s, int which, size_t* len);   /* 0..5, see ef_sched.c */
size_t ef_session_n_ests(const ef_session* s);
void ef_session_close(ef_session* s);

/* environment: PINTRON_THREADS (workers; default: usable cores (affinity, cgroup quota,
 * per local rank), at most 16), PINTRON_LANES (8),
 * PINTRON_FIBERS (fibres per worker over all lanes: 768, or 1024 for reads shorter than 300 bases), PINTRON_FIBER_STACK_KB (256),
 * PINTRON_SERVICES (GPU service threads, 4), PINTRON_COALESCE_US (0), PINTRON_GPU_DEVICE (0), PINTRON_NO_PREFETCH,
 * PINTRON_KERNEL_TIMING, PINTRON_VERBOSE */
/* est-fact over several GPUs of one node from the C program itself (ef_multi.c): `--gpus=N` (or
 * PINTRON_GPUS=N) starts one process per GPU; the ESTs of the gene are split in contiguous ranges,
 * the text of the six files is gathered to rank 0 through pgpu_gather (RCCL) and written there.
 * `--genes=FILE` (directories, one per line) runs many genes in one invocation, gene g on rank
 * g mod N, each leaving its files in its own directory. */
int ef_main_multi(int argc, char** argv);
struct pgpu_ctx* ef_session_context(ef_session* s);
const void* ef_session_genomic(ef_session* s);       /* the genomic record (const ef_seq*, estfact.h) of the session */

/* set by a program that ends right after ef_run_batched: the session is not taken apart */
extern int ef_leave_without_cleanup;
int ef_run_batched(int argc, char** argv);
int ef_run_batched_stats(int argc, char** argv, ef_sched_stats* stats);

#endif
