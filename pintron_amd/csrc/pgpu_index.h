// Internal view of the genomic index (pgpu_index.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/pintron_gpu.h"

const uint8_t* pgpu_index_genomic(const pgpu_index* idx);   // device pointer
size_t pgpu_index_length(const pgpu_index* idx);

// What the suffix-array form of find_longest_common_factor_dp needs (pgpu_dp_kernels.hip: lcfsa_wave_body):
//   focc   first occurrence of every upper-case ACGT l-mer, l = 1..8 (table l at offset (4^l - 4) / 3;
//          0xFFFFFFFF: the l-mer does not occur)
//   rmq    sparse table of range minima over the suffix array: level j >= 1 at (j-1) * n holds
//          min(sa[k .. k + 2^j)) for k + 2^j <= n; level 0 is the suffix array itself
//   first_bad  position of the first character of the sequence that is not an upper-case A, C, G or T
//          (n when there is none): prefixes up to there can be searched with exact matching alone
struct LcfIndexView {
  const uint8_t* T; const uint32_t* sa; const uint32_t* klo; const uint32_t* khi; const uint32_t* focc; const uint32_t* rmq;
  uint32_t n, levels, first_bad;
};
LcfIndexView pgpu_index_lcf_view(const pgpu_index* idx);
constexpr uint32_t LCF_FOCC_ENTRIES = 87380;      // 4 + 16 + ... + 4^8

// context helpers implemented in pgpu_api.hip
hipStream_t pgpu_ctx_stream(pgpu_ctx* ctx);
// makes the context's device current on the calling thread (HIP's current device is per thread:
// a fresh prefetch or service thread starts on device 0)
int pgpu_ctx_bind(pgpu_ctx* ctx);
// waits for everything queued on the context's stream WITHOUT spinning: HIP's own stream synchronisation
// busy-waits, and the thread that drives the pairing / MEG prefetch would burn a core of the host's CPU quota
// for the tens of milliseconds per step its kernels run (PGPU_WAIT: the nap in microseconds, as for DP plans)
hipError_t pgpu_ctx_wait(pgpu_ctx* ctx);
int pgpu_ctx_fail(pgpu_ctx* ctx, int code, const char* msg);
bool pgpu_ctx_pool_acquire(pgpu_ctx* ctx, int pool);
void pgpu_ctx_pool_release(pgpu_ctx* ctx, int pool);
void* pgpu_ctx_pool_get(pgpu_ctx* ctx, int pool, int slot, size_t bytes);
bool pgpu_ctx_pool_share(pgpu_ctx* ctx, int pool);
void pgpu_ctx_pool_unshare(pgpu_ctx* ctx, int pool, const void* who);
void pgpu_ctx_pool_set_owner(pgpu_ctx* ctx, int pool, const void* who);
const void* pgpu_ctx_pool_owner(const pgpu_ctx* ctx, int pool);
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
// PGPU_TRACE_ALLOC=1: every page-locked host allocation of the library on stderr (size, address, call site)
#include <cstdio>
#include <cstdlib>
static inline void pgpu_trace_alloc(const char* what, const void* q, size_t bytes) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("PGPU_TRACE_ALLOC"); on = (e && *e && *e != '0') ? 1 : 0; }
  if (on) fprintf(stderr, "* alloc %-18s %10.1f MB at %p\n", what, bytes / 1048576.0, q);
}

