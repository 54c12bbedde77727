// Internal view of the genomic index (pgpu_index.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/pintron_gpu.h"

const uint8_t* pgpu_index_genomic(const pgpu_index* idx);   // device pointer
size_t pgpu_index_length(const pgpu_index* idx);

// context helpers implemented in pgpu_api.hip
hipStream_t pgpu_ctx_stream(pgpu_ctx* ctx);
// makes the context's device current on the calling thread (HIP's current device is per thread:
// a fresh prefetch or service thread starts on device 0)
int pgpu_ctx_bind(pgpu_ctx* ctx);
int pgpu_ctx_fail(pgpu_ctx* ctx, int code, const char* msg);
bool pgpu_ctx_pool_acquire(pgpu_ctx* ctx, int pool);
void pgpu_ctx_pool_release(pgpu_ctx* ctx, int pool);
void* pgpu_ctx_pool_get(pgpu_ctx* ctx, int pool, int slot, size_t bytes);
bool pgpu_ctx_timing(const pgpu_ctx* ctx);

// pgpu_meg.hip: hand-written exclusive prefix sums and the per-pattern MEG kernels
size_t pgpu_scan_tmp_bytes(size_t n);
void pgpu_exclusive_scan_u32(const uint32_t* in, unsigned long long* out, size_t n, void* tmp, hipStream_t st);
size_t pgpu_meg_scratch_bytes(size_t n_pat);
void pgpu_meg_launch_build(const pgpu_pairing* pairs, const unsigned long long* pair_first, const unsigned long long* pat_off,
                           uint32_t n_pat, const pgpu_meg_params* prm, void* scratch, void* info, uint32_t* rec_bytes,
                           hipStream_t st);
void pgpu_meg_launch_emit(uint32_t n_pat, const void* scratch, const void* info, const unsigned long long* rec_off,
                          uint8_t* out, hipStream_t st);
