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
