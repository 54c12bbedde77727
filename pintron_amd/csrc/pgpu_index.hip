// Genomic index resident in HBM: the sequence (1 B/base) and its suffix array.
#include <new>
#include "pgpu_index.h"

struct pgpu_index {
  uint8_t* d_gen = nullptr;
  uint32_t* d_sa = nullptr;
  size_t len = 0;
};

const uint8_t* pgpu_index_genomic(const pgpu_index* idx) { return idx->d_gen; }
size_t pgpu_index_length(const pgpu_index* idx) { return idx->len; }

extern "C" int pgpu_index_build(pgpu_ctx* ctx, const char* genomic, size_t len, pgpu_index** out) {
  if (!ctx || !out || (len && !genomic)) return PGPU_EINVAL;
  *out = nullptr;
  if (len >= (1u << 28)) return pgpu_ctx_fail(ctx, PGPU_ERANGE, "genomic longer than 2^28");
  pgpu_index* idx = new (std::nothrow) pgpu_index();
  if (!idx) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of host memory");
  idx->len = len;
  hipStream_t st = pgpu_ctx_stream(ctx);
  if (hipMalloc(&idx->d_gen, len + 64) != hipSuccess) { delete idx; return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "hipMalloc genomic"); }
  if (hipMemsetAsync(idx->d_gen, 0, len + 64, st) != hipSuccess ||
      (len && hipMemcpyAsync(idx->d_gen, genomic, len, hipMemcpyHostToDevice, st) != hipSuccess) ||
      hipStreamSynchronize(st) != hipSuccess) {
    hipFree(idx->d_gen); delete idx;
    return pgpu_ctx_fail(ctx, PGPU_EDEVICE, "genomic upload failed");
  }
  *out = idx;
  return PGPU_OK;
}

extern "C" int pgpu_index_destroy(pgpu_ctx* ctx, pgpu_index* idx) {
  if (!ctx || !idx) return PGPU_EINVAL;
  hipStreamSynchronize(pgpu_ctx_stream(ctx));
  hipFree(idx->d_gen); hipFree(idx->d_sa);
  delete idx;
  return PGPU_OK;
}

extern "C" int pgpu_index_suffix_array(pgpu_ctx* ctx, const pgpu_index* idx, uint32_t* sa_out, size_t cap) {
  (void)idx; (void)sa_out; (void)cap;
  return pgpu_ctx_fail(ctx, PGPU_ENOSYS, "suffix array not built in this build");
}

extern "C" int pgpu_pairings(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns,
                             const uint64_t* pat_off, size_t n_pat, const pgpu_pairing_params* params,
                             pgpu_pairing* out, size_t out_cap, uint64_t* out_first, size_t* n_out) {
  (void)idx; (void)patterns; (void)pat_off; (void)n_pat; (void)params; (void)out; (void)out_cap; (void)out_first; (void)n_out;
  return pgpu_ctx_fail(ctx, PGPU_ENOSYS, "pairings not implemented in this build");
}
