// Internal declarations shared by the HIP translation units of libpintron_gpu.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pintron_gpu.h"
#include "pgpu_index.h"

// Job descriptor as the kernels see it: operand pointers already resolved to HBM addresses.
struct DevJob {
  const uint8_t* a;
  const uint8_t* b;
  uint32_t la, lb;
  uint32_t p0, p1, p2, tail;
  uint64_t ws_off;      // byte offset of this job's traceback workspace
  uint64_t str_off;     // byte offset of this job's two alignment strings
  uint32_t out_idx;     // index of the caller's job (results are written in caller order)
  uint32_t r_class;     // rows per lane (R) of the kernel instance that ran the job
};
static_assert(sizeof(DevJob) == 64, "DevJob layout");

using DevResult = pgpu_dp_result;

// kernel families (one launch group per family and row class)
enum KernelFamily {
  KF_ALIGN = 0, KF_GAP, KF_ED, KF_KBAND, KF_LCF, KF_BORDERS, KF_AFFIX,
  KF_LCFSA,            // find_longest_common_factor_dp answered from the suffix array (one wave per job; see lcfsa_wave_body)
  KF_LCFW,             // ... of two short strings (the small-exon search's exon ends): one wave per job, a lane per diagonal
  KF_ALIGNB,           // ALIGN above 64 rows whose lengths differ by at most the band's half-width: inside the band on ONE
                       // wave (align_band_sweep); a job whose banded score exceeds the band is left with status
                       // ALIGN_BAND_RETRY and finished by the follow-up launch (launch_align_fallback) on four waves
  KF_COUNT
};

// launchers (pgpu_dp_kernels.hip)
void launch_lev(int mode_family, int R, uint32_t max_rows, const DevJob* jobs, int njobs, int n_big, DevResult* res,
                uint8_t* ws, uint8_t* strs, hipStream_t st);     // ALIGN: matrix + traceback
void launch_gap(const DevJob* jobs, int njobs, int n_big, DevResult* res, uint8_t* ws, uint8_t* strs, hipStream_t st);   // + traceback
void launch_gap_slow(const DevJob* jobs, int njobs, DevResult* res, uint8_t* ws, uint8_t* strs, hipStream_t st);   // beyond 2048 rows
constexpr int MAX_WAVE_SEGS = 10;
constexpr int32_t ALIGN_BAND_RETRY = 1;     // DevResult.status of a banded ALIGN that has to be swept in full (never leaves the library)
constexpr uint32_t ALIGN_BAND_HALF = 31u;   // half-width of the band (2k + 1 <= 64 lanes)
// the whole-matrix sweep (four waves per job) for the jobs of [jobs, jobs + njobs) the band could not settle
void launch_align_fallback(const DevJob* jobs, int njobs, DevResult* res, uint8_t* ws, uint8_t* strs, hipStream_t st);
void launch_wave_jobs(const DevJob* jobs, int n_segs, const int* family, const int* start, const int* count,
                      DevResult* res, uint8_t* ws, uint8_t* strs, const LcfIndexView& ix, hipStream_t st);   // every wave-per-job family in one launch
// one launch for the one-job-per-workgroup sweeps and the wave-per-job families of a batch; returns
// false (nothing launched) when the largest BORDERS pattern needs more LDS than a workgroup may share
bool launch_dp_batch(const DevJob* jobs, int n_segs, const int* family, const int* start, const int* count,
                     int bc_start, int bc_count, uint32_t bc_max_rows, int ac_start, int ac_count, int lc_start, int lc_count,
                     DevResult* res, uint8_t* ws, uint8_t* strs, const LcfIndexView& ix, hipStream_t st);
size_t dp_batch_lds_bytes(bool wave_jobs, int bc_count, uint32_t bc_max_rows, int ac_count, int lc_count);
// ALIGN with 65 .. 4096 rows runs on four waves (align_coop_body): rows per lane of the 256-lane sweep, and the
// bytes of one traceback entry (all rows of one lane in one column)
static inline uint32_t align_coop_entry_bytes(uint32_t r_class) { const uint32_t R = r_class <= 4 ? 1u : r_class / 4; return R <= 4 ? 1u : R / 4; }
// keys: one zeroed entry per job, (length << 44) | (2^28-1 - occ1) << 16 | (2^16-1 - occ2) of the best run; 0: none
void launch_lcf(const DevJob* jobs, int njobs, uint32_t max_chunks, uint32_t max_l2,
                unsigned long long* keys, hipStream_t st);

// bytes of one traceback entry (all rows of one lane in one column)
// row class of jobs with more than 4096 rows: run by the R = 64 kernels in strips of 4096 rows
constexpr uint32_t ROW_CLASS_STRIPS = 128;
// one boundary row (the last row of a strip, per column) in the job's workspace; there are two
__host__ __device__ inline size_t strip_bnd_bytes(uint32_t nc) { return (((size_t)nc + 1) * 4 + 15) & ~(size_t)15; }
static inline uint32_t align_entry_bytes(uint32_t R) { return R == ROW_CLASS_STRIPS ? 16u : (R <= 4 ? 1u : R / 4); }
static inline uint32_t gap_entry_bytes(uint32_t R) { return R; }

