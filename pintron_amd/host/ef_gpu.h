/* GPU backend of the est-fact host program (over include/pintron_gpu.h). */
#ifndef EF_GPU_H
#define EF_GPU_H

#include "estfact.h"
#include "../../include/pintron_gpu.h"

/* growable (job table, operand arena) pair in the C-ABI's own layout */
typedef struct {
  pgpu_dp_job* jobs; size_t n, cap;
  char* arena; size_t arena_len, arena_cap;
} ef_jobbuf;

void ef_jobbuf_init(ef_jobbuf* jb);
void ef_jobbuf_reset(ef_jobbuf* jb);
void ef_jobbuf_free(ef_jobbuf* jb);
size_t ef_jobbuf_add(ef_jobbuf* jb, const ef_dp_req* q, const char* gen, size_t gen_len);
int ef_decode_result(int kind, const pgpu_dp_result* r, const char* strings, ef_dp_res* out);
/* PINTRON_DP_TRACE=<file>: requests and answers for tools/replay_dp_trace.py (ef_gpu_backend.c) */
int ef_dp_trace_enabled(void);
void ef_dp_trace(const ef_dp_req* q, const ef_dp_res* r, uint32_t unit);
void ef_dp_trace_flush(void);
int ef_gpu_device_from_env(void);          /* PINTRON_GPU_DEVICE, default 0 */

/* direct mode: one C-ABI call per request */
ef_backend* ef_gpu_open(const ef_seq* gen);
void ef_gpu_close(ef_backend* be);

#endif
