// Hand-written gfx950 (CDNA4, wave64) kernels for the est-fact dynamic programs.
//
// Common structure of the Levenshtein-family and gap kernels ("row strips x skewed column sweep"):
//   * one wavefront (64 lanes) owns one job; a 256-thread workgroup carries four independent jobs;
//   * lane l keeps R consecutive DP rows (l*R+1 .. l*R+R) of the CURRENT column in registers
//     (R is a template parameter: 1,2,4,...,64 -> up to 4096 rows);
//   * the wave sweeps the columns with a skew of one column per lane: at step s lane l works on
//     column s-l+1, so the anti-diagonal of strips is processed in parallel;
//   * the only inter-lane traffic per step is ONE 32-bit DPP wave shift (wave_shr:1) that hands
//     the value of the strip's last row -- packed with the column character, which therefore
//     travels down the lanes with the wavefront instead of being re-read -- to the lane below;
//   * traceback directions are packed (2 bits/cell, or 5 bits/cell for the 3-plane gap DP) and
//     stored step-major, [step][lane], so that every store instruction of the wave writes one
//     contiguous 64*entry-byte segment of HBM;
//   * tracebacks run in a second kernel, one wave per job: the chain of dependent direction
//     look-ups is walked in LDS (windows of the direction stream are staged there) with the walk
//     state in scalar registers, and the gapped strings are written by all lanes at once.
//
// Integer/character work only: no MFMA.  Reference routines are cited per kernel; paths are
// relative to the AlgoLab/PIntron tree.
#include "pgpu_internal.h"
#include <stdlib.h>

namespace {

constexpr uint32_t PAD_ROW = 0x01u;   // never equal to a sequence byte nor to PAD_COL
constexpr uint32_t PAD_COL = 0x02u;

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {
  // DPP wave_shr:1 -- lane l receives lane l-1's v; lane 0 keeps its own (overwritten by caller)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) {
  // DPP wave_shl:1 -- lane l receives lane l+1's v; lane 63 keeps its own (the caller masks it)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false);
}
// lane l receives lane l-1's v, lane 0 receives ITS OWN `first` (a lane without a source keeps the old value)
__device__ __forceinline__ uint32_t wave_shr1_first(uint32_t first, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
}
// lane l receives lane l+1's v, lane 63 receives its own `last`
__device__ __forceinline__ uint32_t wave_shl1_last(uint32_t last, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130, 0xf, 0xf, false);
}

__device__ __forceinline__ bool is_n(uint32_t c) { return c == 'n' || c == 'N'; }

// getBursetFrequency (src/refine-intron.c:376-556) as a table: index = donor[0],donor[1],
// acceptor[0],acceptor[1] at 2 bits each (A=0,C=1,G=2,T=3).
__constant__ uint8_t c_burset[256] = {
    0,   0,   1,   1,   0,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,
    0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   1,   5,   0,   0,   0,   0,   2,   0,   1,   0,   0,   0,   0,   2,   0,
    1,   8,   7,   2,   0,   0,   0,   0,   0,   1,   0,   1,   0,   0,   0,   0,
    0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   1,
    0,   0,   2,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   0,   1,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   2,   0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   0,   8,   0,   0,   0,   0,   0,   0,   0,   0,   1,   0,   1,   1,   0,
    0,   0, 126,   0,   0,   0,   0,   0,   0,   0,   1,   0,   1,   0,   0,   0,
    0,   1,  11,   0,   1,   0,   0,   0,   2,   0,   0,   0,   0,   2,   0,   0,
    0,   4, 200,   2,   9,   0,   4,   3,   0,   1,  10,   1,   7,   2,   8,   2,
    0,   0,   6,   0,   0,   0,   1,   0,   0,   0,   0,   0,   0,   1,   0,   0,
    0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,
    0,   1,   7,   0,   0,   0,   0,   0,   0,   0,   2,   0,   0,   0,   0,   0,
    0,   0,   5,   1,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,
};

__device__ __forceinline__ int base_code(uint32_t c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

// getBursetFrequency_adaptor (src/refine-intron.c:362-374) over t with `avail` readable bytes
__device__ int burset_adaptor(const uint8_t* t, uint32_t avail, uint32_t cut1, uint32_t cut2) {
  if (cut2 < 2) return 0;
  if (cut1 + 1 >= avail || cut2 - 1 >= avail) return 0;   // a NUL terminates the C string
  const int c0 = base_code(t[cut1]), c1 = base_code(t[cut1 + 1]);
  const int c2 = base_code(t[cut2 - 2]), c3 = base_code(t[cut2 - 1]);
  if ((c0 | c1 | c2 | c3) < 0) return 0;
  return c_burset[(c0 << 6) | (c1 << 4) | (c2 << 2) | c3];
}

// ---------------------------------------------------------------------------------------------
// Levenshtein family
// ---------------------------------------------------------------------------------------------

struct Operand {            // a string read forwards or backwards
  const uint8_t* base;
  uint32_t total;           // length of the underlying buffer (for reversed reads)
  bool rev;
  __device__ __forceinline__ uint32_t at(uint32_t i) const {
    return rev ? base[total - 1 - i] : base[i];
  }
};

template <int R> struct DirPack {          // 2 bits per row, R rows
  static constexpr int WORDS = (R + 15) / 16;
  uint32_t w[WORDS];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < WORDS; ++i) w[i] = 0;
  }
  __device__ __forceinline__ void set(int r, uint32_t d) { w[r / 16] |= d << (2 * (r % 16)); }
  __device__ __forceinline__ void store(uint8_t* p) const {
    if constexpr (R <= 4)       *p = (uint8_t)w[0];
    else if constexpr (R == 8)  *reinterpret_cast<uint16_t*>(p) = (uint16_t)w[0];
    else if constexpr (R == 16) *reinterpret_cast<uint32_t*>(p) = w[0];
    else if constexpr (R == 32) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
    else                        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

struct AffixBest {          // running best cut of find_longest_affix; s == 0: none yet
  uint32_t v, s;            // its distance and e + g
  uint64_t key;             // (e << 32) | g: the cell scanned later has the larger key
  __device__ __forceinline__ uint32_t valid() const { return s != 0u ? 1u : 0u; }
  __device__ __forceinline__ uint32_t e() const { return (uint32_t)(key >> 32); }
  __device__ __forceinline__ uint32_t g() const { return (uint32_t)key; }
  // One cell of the scan: the cell is a candidate when the characters match and
  // 200*v <= 17*(e+g); it replaces the running best when its weight v/(e+g) is smaller, or equal
  // with a later (e,g).  `small`: e+g < 2^15, so every product fits 24x24 -> 32 bits (full-rate
  // v_mul_u32_u24); otherwise 64-bit products.  With no best yet (v = 1, s = 0) the first
  // candidate wins: cv * 0 < 1 * cs.
  template <bool SMALL>
  __device__ __forceinline__ void consider(bool match, uint32_t cv, uint32_t ce, uint32_t cg) {
    const uint32_t cs = ce + cg;
    bool cand, less, equal;
    if constexpr (SMALL) cand = match & (__umul24(200u, cv) <= __umul24(17u, cs));
    else                 cand = match & (200ull * cv <= 17ull * cs);
    // one scalar branch skips the comparison against the running best when no lane holds a candidate
    if (!__builtin_amdgcn_ballot_w64(cand)) return;
    if constexpr (SMALL) {
      const uint32_t lhs = __umul24(cv, s), rhs = __umul24(v, cs);
      less = lhs < rhs; equal = lhs == rhs;
    } else {
      const uint64_t lhs = (uint64_t)cv * s, rhs = (uint64_t)v * cs;
      less = lhs < rhs; equal = lhs == rhs;
    }
    const uint64_t ck = ((uint64_t)ce << 32) | cg;
    const bool take = cand & (less | (equal & (ck > key)));
    v = take ? cv : v; s = take ? cs : s; key = take ? ck : key;
  }
  // true when candidate o replaces this one under the reference's scan rule: smaller weight wins,
  // equal weight -> the cell scanned later (larger (e,g)) wins.
  __device__ __forceinline__ bool worse_than(const AffixBest& o) const {
    if (!s) return true;
    const uint64_t lhs = (uint64_t)o.v * s, rhs = (uint64_t)v * o.s;
    if (lhs != rhs) return lhs < rhs;
    return o.key > key;
  }
  __device__ __forceinline__ AffixBest from_lane_xor(int off) const {
    AffixBest o;
    o.v = __shfl_xor(v, off); o.s = __shfl_xor(s, off);
    o.key = ((uint64_t)__shfl_xor((uint32_t)(key >> 32), off) << 32) | __shfl_xor((uint32_t)key, off);
    return o;
  }
};
constexpr AffixBest AFFIX_NONE{1u, 0u, 0ull};

// One column sweep.  On return cur[r] = M[row(l,r)][nc].
constexpr uint32_t BAND_INF = 0x3FFFFFu;   // "outside the band"; stays below the 24-bit value field

// Rows beyond 64*R are processed in horizontal STRIPS of 64*R rows by the same wave: the strip's
// last row is written, per column, to a boundary array in the job's workspace (`bottom`) and is
// the row above the first row (`top`) of the next strip; `row_base` = rows before this strip.
// The boundary values go through memory written and read by one wave: agent-scope atomics keep
// the per-CU L1 out of the way.
template <int R, bool WILD, bool DIRS, bool ROWMIN, bool AFFIX, bool BAND = false, bool ASMALL = false>
__device__ __forceinline__ void lev_sweep_strip(const Operand rows, const uint32_t nr,
                                          const Operand cols, const uint32_t nc,
                                          const uint32_t lane, uint32_t (&cur)[R],
                                          uint32_t (&minv)[R], uint32_t (&minpos)[R],
                                          AffixBest& best, uint8_t* dir_ws, const uint32_t band_k = 0,
                                          const uint32_t row_base = 0, const uint32_t* top_row = nullptr,
                                          uint32_t* bottom_row = nullptr) {
  uint32_t rc[R];                       // row characters of this lane's strip
  const uint32_t row0 = lane * R;       // rows row0+1 .. row0+R (of the strip)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = row0 + r;
    rc[r] = i < nr ? rows.at(i) : PAD_ROW;
    cur[r] = row_base + i + 1;          // M[i+1][0]
    if constexpr (BAND) { if (row_base + i + 1 > band_k) cur[r] = BAND_INF; }
    if constexpr (ROWMIN) { minv[r] = i + 1; minpos[r] = 0; }
  }
  if (nr == 0 || nc == 0) return;
  const uint32_t last_lane = (nr - 1) / R;
  const uint32_t last_r = (nr - 1) % R;
  const uint32_t steps = nc + last_lane;
  uint32_t diag_in = row_base + row0;   // M[row0][j-1] for j = 1
  uint32_t out = 0;                     // (value of the strip's last row) | (column char << 24)
  uint32_t chunk = 0, tchunk = 0;
  constexpr uint32_t EB = R <= 4 ? 1u : R / 4;
  if (bottom_row && lane == 0) {
    uint32_t b0 = row_base + nr;
    if constexpr (BAND) { if (b0 > band_k) b0 = BAND_INF; }
    __hip_atomic_store(&bottom_row[0], b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (top_row && lane == 0) diag_in = __hip_atomic_load(&top_row[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  for (uint32_t s = 0; s < steps; ++s) {
    const uint32_t t = s & 63u;
    if (t == 0) {                       // refill the column-character window (coalesced 64 B)
      const uint32_t j = s + lane;
      chunk = j < nc ? cols.at(j) : PAD_COL;
      if (top_row) tchunk = j < nc ? __hip_atomic_load(&top_row[j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    uint32_t in = wave_shr1(out);
    const uint32_t ch0 = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)t);
    const uint32_t top_in = (uint32_t)__builtin_amdgcn_readlane((int)tchunk, (int)t);
    if (lane == 0) {
      uint32_t top = top_row ? top_in : s + 1;             // M[row_base][j], j = s+1
      if constexpr (BAND) { if (!top_row && top > band_k) top = BAND_INF; }
      in = top | (ch0 << 24);
    }
    const uint32_t j = s - lane + 1;                     // column of this lane (wraps when idle)
    if (j - 1u < nc) {
      const uint32_t ch = in >> 24;
      const uint32_t in_val = in & 0xFFFFFFu;
      uint32_t up = in_val;
      uint32_t diag = diag_in;
      const bool ch_n = WILD && is_n(ch);
      DirPack<R> dp;
      if constexpr (DIRS) dp.clear();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t left = cur[r];
        bool match = rc[r] == ch;
        if constexpr (WILD) match = match || ch_n || is_n(rc[r]);
        uint32_t v = diag + (match ? 0u : 1u);
        if constexpr (DIRS) {
          // ComputeAlignMatrix tie-break: diagonal, then up (dir 1), then left (dir 2), strict >
          uint32_t d = 0;
          if (v > up + 1) { v = up + 1; d = 1; }
          if (v > left + 1) { v = left + 1; d = 2; }
          dp.set(r, d);
        } else {
          v = min(v, min(up + 1, left + 1));
        }
        if constexpr (BAND) {
          // K_band_edit_distance keeps cells with |column - row| <= k only; neighbours outside
          // the band do not take part in the minimum (src/compute-alignments.c:375-443)
          const uint32_t row = row_base + row0 + r + 1;
          v = (j + band_k >= row && row + band_k >= j) ? min(v, BAND_INF) : BAND_INF;
        }
        if constexpr (ROWMIN) {
          if (minv[r] > v) { minv[r] = v; minpos[r] = j; }   // strict: first arg-min
        }
        if constexpr (AFFIX) {
          // cut_weight = 2*v/(e+g) <= 0.17  <=>  200*v <= 17*(e+g)   (exact)
          best.template consider<ASMALL>(rc[r] == ch, v, row_base + row0 + r + 1, j);
        }
        diag = left;
        cur[r] = v;
        up = v;
      }
      diag_in = in_val;
      out = up | (ch << 24);
      if constexpr (DIRS) dp.store(dir_ws + ((size_t)s * 64 + lane) * EB);
      if (bottom_row && lane == last_lane) {
        uint32_t bv = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) if ((uint32_t)r == last_r) bv = cur[r];
        __hip_atomic_store(&bottom_row[j], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// One column sweep over at most 64*R rows (the common case; strips of longer jobs: lev_sweep_strip).
// The step loop is a chain of dependent instructions on one wave, so each instruction in it is
// latency: per 64-step chunk the inputs of lane 0 (top value | column character) are prepared by
// all lanes at once in ONE register (`feed`, lane t = step t) that moves one lane down per step, so
// lane 0 always holds the input of the current step and the DPP shift that passes the strips' last
// rows along drops it in: two DPP moves per step, no scalar round trip.  On return
// cur[r] = M[row(l,r)][nc].
template <int R, bool WILD, bool DIRS, bool ROWMIN, bool AFFIX, bool BAND = false, bool ASMALL = false>
__device__ __forceinline__ void lev_sweep(const Operand rows, const uint32_t nr,
                                          const Operand cols, const uint32_t nc,
                                          const uint32_t lane, uint32_t (&cur)[R],
                                          uint32_t (&minv)[R], uint32_t (&minpos)[R],
                                          AffixBest& best, uint8_t* dir_ws, const uint32_t band_k = 0) {
  uint32_t rc[R];                       // row characters of this lane's strip
  const uint32_t row0 = lane * R;       // rows row0+1 .. row0+R
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = row0 + r;
    rc[r] = i < nr ? rows.at(i) : PAD_ROW;
    cur[r] = i + 1;                     // M[i+1][0]
    if constexpr (BAND) { if (i + 1 > band_k) cur[r] = BAND_INF; }
    if constexpr (ROWMIN) { minv[r] = i + 1; minpos[r] = 0; }
  }
  if (nr == 0 || nc == 0) return;
  const uint32_t last_lane = (nr - 1) / R;
  const uint32_t steps = nc + last_lane;
  uint32_t diag_in = row0;              // M[row0][j-1] for j = 1
  uint32_t out = 0;                     // (value of the strip's last row) | (column char << 24)
  constexpr uint32_t EB = R <= 4 ? 1u : R / 4;

  // the column characters of chunk c+1 are requested while chunk c is swept (the load would otherwise sit
  // at the head of every chunk's dependent chain: ~1 us of L2/HBM latency per 64 steps)
  auto feed_of = [&](const uint32_t s0) -> uint32_t {
    const uint32_t jc = s0 + lane;      // column jc+1 enters lane 0 at step jc
    uint32_t top = jc + 1u;             // M[0][jc+1]
    if constexpr (BAND) { if (top > band_k) top = BAND_INF; }
    return top | ((jc < nc ? cols.at(jc) : PAD_COL) << 24);
  };
  uint32_t feed_next = feed_of(0);
  for (uint32_t s0 = 0; s0 < steps; s0 += 64) {
    uint32_t feed = feed_next;
    feed_next = feed_of(s0 + 64u);
    const uint32_t tmax = (min(64u, steps - s0) + 7u) & ~7u;   // whole groups of 8; steps past the end touch no cell
    for (uint32_t t0 = 0; t0 < tmax; t0 += 8) {
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) {
        const uint32_t s = s0 + t0 + u;
        const uint32_t in = wave_shr1_first(feed, out);
        feed = wave_shl1(feed);
        const uint32_t j = s - lane + 1;                   // column of this lane (wraps when idle)
        if (j - 1u < nc) {
          const uint32_t ch = in >> 24;
          const uint32_t in_val = in & 0xFFFFFFu;
          uint32_t up = in_val;
          uint32_t diag = diag_in;
          const bool ch_n = WILD && is_n(ch);
          DirPack<R> dp;
          if constexpr (DIRS) dp.clear();
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t left = cur[r];
            bool match = rc[r] == ch;
            if constexpr (WILD) match = match || ch_n || is_n(rc[r]);
            uint32_t v = diag + (match ? 0u : 1u);
            if constexpr (DIRS) {
              // ComputeAlignMatrix tie-break: diagonal, then up (dir 1), then left (dir 2), strict >
              uint32_t d = 0;
              if (v > up + 1) { v = up + 1; d = 1; }
              if (v > left + 1) { v = left + 1; d = 2; }
              dp.set(r, d);
            } else {
              v = min(v, min(up + 1, left + 1));
            }
            if constexpr (BAND) {
              // K_band_edit_distance keeps cells with |column - row| <= k only; neighbours outside
              // the band do not take part in the minimum (src/compute-alignments.c:375-443)
              const uint32_t row = row0 + r + 1;
              v = (j + band_k >= row && row + band_k >= j) ? min(v, BAND_INF) : BAND_INF;
            }
            if constexpr (ROWMIN) {
              if (minv[r] > v) { minv[r] = v; minpos[r] = j; }   // strict: first arg-min
            }
            if constexpr (AFFIX) {
              // cut_weight = 2*v/(e+g) <= 0.17  <=>  200*v <= 17*(e+g)   (exact)
              best.template consider<ASMALL>(rc[r] == ch, v, row0 + r + 1, j);
            }
            diag = left;
            cur[r] = v;
            up = v;
          }
          diag_in = in_val;
          out = up | (ch << 24);
          if constexpr (DIRS) dp.store(dir_ws + ((size_t)s * 64 + lane) * EB);
        }
      }
    }
  }
}

// value of row `row` (1-based) after a sweep, written by the lane that owns it (predicated
// stores instead of a dynamically indexed register array)
template <int R>
__device__ __forceinline__ void store_row_value(const uint32_t (&a)[R], uint32_t lane, uint32_t row,
                                                int32_t* dst) {
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (lane * R + r + 1 == row) *dst = (int32_t)a[r];
}

enum { MODE_ED = 0, MODE_ALIGN = 1, MODE_BORDERS = 2, MODE_AFFIX = 3, MODE_KBAND = 4 };

// MODE_ED      edit_distance last cell (src/refine.c:50-83) / compute_edit_distance
//              (src/compute-alignments.c:240-249)
// MODE_ALIGN   ComputeAlignMatrix (src/compute-alignments.c:85-147) incl. the equal-string
//              shortcut of compute_alignment (:48-58); directions go to the workspace
// MODE_BORDERS general_refine_borders (src/refine.c:105-192)
// MODE_AFFIX   find_longest_affix (src/factorization-refinement.c:1136-1173)
// STRIPS (R = 64 only): jobs with more than 4096 rows, swept in strips of 4096 rows; the job's
// workspace starts with the two boundary rows (strip_bnd_bytes each), ALIGN directions follow,
// one block of (columns + 64) * 64 * 16 bytes per strip.

// dustScore (src/exon-complexity.c:50-78) of s[0..len) on one wave: every dinucleotide adds the number of times it has
// been seen before, i.e. the sum over the 17 dinucleotide classes (getDinucleotideIndex :80-130: A, C, G, T in either
// case, everything else class 16) of f (f - 1) / 2 for their final counts f -- an integer, so the FP64 arithmetic that
// follows (x 10.0, / (length - 2), / length) sees the reference's running total.  The counts come from ballots over 64
// positions at a time.  Same value in every lane.
__device__ __noinline__ double dust_score_wave(const uint8_t* __restrict__ s, const uint32_t len, const uint32_t lane) {
  if ((int)len <= 2) return 0.0;
  uint32_t cnt[17];
#pragma unroll
  for (int b = 0; b < 17; ++b) cnt[b] = 0u;
  const uint32_t nd = len - 1u;                  // dinucleotides
  for (uint32_t base = 0; base < nd; base += 64u) {
    const uint32_t i = base + lane;
    int cls = -1;                                // no dinucleotide on this lane
    if (i < nd) {
      const int x = base_code(s[i]), y = base_code(s[i + 1]);
      cls = (x < 0 || y < 0) ? 16 : 4 * x + y;
    }
#pragma unroll
    for (int b = 0; b < 17; ++b) cnt[b] += (uint32_t)__popcll(__ballot(cls == b));
  }
  unsigned long long running = 0ull;
#pragma unroll
  for (int b = 0; b < 17; ++b) running += (unsigned long long)cnt[b] * (cnt[b] - (cnt[b] ? 1u : 0u)) / 2ull;
  const double dust = (10.0 * (double)running) / ((double)(len - 2u));
  return dust / (double)len;
}
// the exon check's flags (pgpu_gpu.h: KBAND with tail = 1): bit 0 dust(a) > threshold, bit 1 dust(b) > threshold
__device__ __forceinline__ uint32_t dust_flags_wave(const DevJob& job, const uint32_t lane) {
  const double thr = __longlong_as_double((long long)(((unsigned long long)job.p2 << 32) | (unsigned long long)job.p1));
  const double da = dust_score_wave(job.a, job.la, lane), db = dust_score_wave(job.b, job.lb, lane);
  return (da > thr ? 1u : 0u) | (db > thr ? 2u : 0u);
}

// K_band_edit_distance (src/compute-alignments.c:375-443) with THE BAND ON THE LANES: lane s owns
// slot s of the reference's 2k+1 wide row buffers, i.e. the diagonal column - row = s - k, and the
// wave walks down the rows.  Cell (r, s) needs (r-1, s) [diagonal: the lane's own previous value],
// (r-1, s+1) [up: the right neighbour's previous row] and (r, s-1) [left: the left neighbour's same
// row], so lane s takes row r at time 2r + s: even lanes in the first half of an iteration, odd
// lanes in the second, each reading its neighbours' latest value over DPP.  m + k iterations of two
// single-cell half-steps instead of (n + 63) steps of R cells on the matrix sweep.  Needs 2k+1 <= 64.
// A lane works on the rows whose column lies in 1..n; before its first row it holds the boundary
// value next to it -- M[0][s-k] = s-k for the slots right of the main diagonal, M[k-s][0] = k-s for
// those left of it -- which is what its right neighbour reads as "left" and itself as "diagonal"
// on its first row.  Slot 0 has no left term and slot 2k no up term, as in the reference's loops.
__device__ __noinline__ void kband_band_sweep(const uint8_t* __restrict__ lng, const uint32_t n,
                                              const uint8_t* __restrict__ sht, const uint32_t m,
                                              const uint32_t k, const uint32_t lane, DevResult* res) {
  const uint32_t W = 2u * k + 1u;
  const bool used = lane < W;
  const bool odd = (lane & 1u) != 0u;
  const int off = (int)lane - (int)k;            // column - row on this lane's diagonal
  const uint32_t half = lane >> 1;               // iteration q works on row r = q - half
  // neighbours that do not exist are pushed out of the minimum
  const uint32_t up_mask = (lane + 1u < W) ? 0u : BAND_INF, left_mask = lane > 0u ? 0u : BAND_INF;
  uint32_t val = (uint32_t)(off < 0 ? -off : off);
  // rows of this lane: 1 <= r <= m with 1 <= off + r <= n
  const uint32_t r_lo = off < 0 ? (uint32_t)(1 - off) : 1u;
  const int hi_i = (int)n - off < (int)m ? (int)n - off : (int)m;
  const uint32_t span = (used && hi_i >= (int)r_lo) ? (uint32_t)hi_i - r_lo : 0xFFFFFFFFu;   // r - r_lo <= span: active
  const bool any_row = used && hi_i >= (int)r_lo;
  const uint32_t nq = m + k;
  // characters of the cell of iteration q (the loads run a group of four iterations ahead)
  auto row_char = [&](uint32_t q) -> uint32_t {
    const uint32_t r = q - half;
    return (any_row && r - r_lo <= span) ? (uint32_t)sht[r - 1u] : 0u;
  };
  auto col_char = [&](uint32_t q) -> uint32_t {
    const uint32_t r = q - half;
    return (any_row && r - r_lo <= span) ? (uint32_t)lng[(uint32_t)(off + (int)r) - 1u] : 1u;
  };
  uint32_t a[4], b[4], an[4], bn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = row_char(1u + j); b[j] = col_char(1u + j); }
  for (uint32_t q0 = 1; q0 <= nq; q0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { an[j] = row_char(q0 + 4u + j); bn[j] = col_char(q0 + 4u + j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t r = q0 + j - half;          // wraps for the lanes whose first row lies ahead
      const bool active = any_row && (r - r_lo <= span);
      const uint32_t mism = a[j] != b[j] ? 1u : 0u;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const uint32_t upv = wave_shl1(val) | up_mask, leftv = wave_shr1(val) | left_mask;
        const uint32_t nv = min(val + mism, min(upv, leftv) + 1u);
        val = (active && odd == (par == 1)) ? nv : val;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = an[j]; b[j] = bn[j]; }
  }
  if (lane == n + k - m) { res->status = 0; res->v[1] = (int32_t)val; res->v[0] = val <= k ? 1 : 0; }
}

template <int R, int MODE, bool STRIPS = false>
__device__ __forceinline__ void lev_wave_body(const DevJob& job, DevResult* res, uint8_t* __restrict__ ws,
                                              const uint32_t lane, uint32_t* wave_lds = nullptr) {
  uint32_t cur[R], minv[R], minpos[R];
  AffixBest best = AFFIX_NONE;

  if constexpr (MODE == MODE_ED) {
    // distance is symmetric: keep the shorter string on the rows
    const bool swap = job.la > job.lb;
    const Operand rows{swap ? job.b : job.a, 0, false}, cols{swap ? job.a : job.b, 0, false};
    const uint32_t nr = swap ? job.lb : job.la, nc = swap ? job.la : job.lb;
    if constexpr (STRIPS) {
      constexpr uint32_t SR = 64u * R;
      uint32_t* bnd = reinterpret_cast<uint32_t*>(ws + job.ws_off);
      const size_t bw = strip_bnd_bytes(nc) / 4;
      uint32_t done = 0;
      for (uint32_t k = 0; done < nr; ++k, done += SR) {
        const uint32_t part = min(SR, nr - done);
        const Operand rs{rows.base + done, 0, false};
        lev_sweep_strip<R, false, false, false, false>(rs, part, cols, nc, lane, cur, minv, minpos, best, nullptr, 0, done,
                                                 k ? bnd + ((k - 1) & 1) * bw : nullptr, done + part < nr ? bnd + (k & 1) * bw : nullptr);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the next strip reads this strip's last row
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      store_row_value<R>(cur, lane, nr - (done - SR), &res->v[0]);
      if (lane == 0) res->status = 0;
      return;
    }
    lev_sweep<R, false, false, false, false>(rows, nr, cols, nc, lane, cur, minv, minpos, best, nullptr);
    if (nr == 0) { if (lane == 0) { res->status = 0; res->v[0] = (int32_t)nc; } return; }
    store_row_value<R>(cur, lane, nr, &res->v[0]);
    if (lane == 0) res->status = 0;
  } else if constexpr (MODE == MODE_ALIGN) {
    const uint32_t n = job.la, m = job.lb;
    bool same = n == m;
    if (same) for (uint32_t i = lane; i < n; i += 64) same = same && job.a[i] == job.b[i];
    if (__all(same)) {                   // identity alignment, score 0 (compute-alignments.c:48-58)
      if (lane == 0) { res->status = 0; res->v[0] = 0; res->v[1] = (int32_t)n; res->v[5] = 1; }
      return;
    }
    const Operand rows{job.a, 0, false}, cols{job.b, 0, false};
    if constexpr (STRIPS) {
      constexpr uint32_t SR = 64u * R;
      uint32_t* bnd = reinterpret_cast<uint32_t*>(ws + job.ws_off);
      const size_t bw = strip_bnd_bytes(m) / 4;
      uint8_t* dirs = ws + job.ws_off + 2 * strip_bnd_bytes(m);
      const size_t strip_dirs = ((size_t)m + 64) * 64 * (R / 4);
      uint32_t done = 0;
      for (uint32_t k = 0; done < n; ++k, done += SR) {
        const uint32_t part = min(SR, n - done);
        const Operand rs{job.a + done, 0, false};
        lev_sweep_strip<R, true, true, false, false>(rs, part, cols, m, lane, cur, minv, minpos, best, dirs + k * strip_dirs, 0, done,
                                               k ? bnd + ((k - 1) & 1) * bw : nullptr, done + part < n ? bnd + (k & 1) * bw : nullptr);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the next strip reads this strip's last row
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      store_row_value<R>(cur, lane, n - (done - SR), &res->v[0]);
      if (lane == 0) { res->status = 0; res->v[5] = 0; }
      return;
    }
    lev_sweep<R, true, true, false, false>(rows, n, cols, m, lane, cur, minv, minpos, best, ws + job.ws_off);
    if (n == 0 || m == 0) { if (lane == 0) { res->status = 0; res->v[0] = (int32_t)(n + m); res->v[5] = 0; } return; }
    store_row_value<R>(cur, lane, n, &res->v[0]);
    if (lane == 0) { res->status = 0; res->v[5] = 0; }
  } else if constexpr (MODE == MODE_BORDERS) {
    // pre[], pre_pos[], suf[], suf_pos[], each len_p+1: the wave's own LDS region when the caller
    // hands one in (several jobs per workgroup), else the workgroup's dynamic LDS
    extern __shared__ uint32_t lds_dyn[];
    uint32_t* lds = wave_lds ? wave_lds : lds_dyn;
    const uint32_t len_p = job.la, len_t = job.lb, max_errs = job.p2;
    const uint32_t t_win = min(len_p + max_errs, len_t);
    uint32_t* pre = lds; uint32_t* pre_pos = pre + (len_p + 1);
    uint32_t* suf = pre_pos + (len_p + 1); uint32_t* suf_pos = suf + (len_p + 1);
    {
      const Operand rows{job.a, len_p, false}, cols{job.b, len_t, false};
      lev_sweep<R, false, false, true, false>(rows, len_p, cols, t_win, lane, cur, minv, minpos, best, nullptr);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t i = lane * R + r + 1;
        if (i <= len_p) { pre[i] = minv[r]; pre_pos[i] = minpos[r]; }
      }
    }
    {
      const Operand rows{job.a, len_p, true}, cols{job.b, len_t, true};
      lev_sweep<R, false, false, true, false>(rows, len_p, cols, t_win, lane, cur, minv, minpos, best, nullptr);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t i = lane * R + r + 1;
        if (i <= len_p) { suf[i] = minv[r]; suf_pos[i] = minpos[r]; }
      }
    }
    if (lane == 0) { pre[0] = 0; pre_pos[0] = 0; suf[0] = 0; suf_pos[0] = 0; }
    // one wave produced the four arrays and one wave reads them: a wave-level hand-over (a workgroup
    // barrier would couple this wave to the unrelated jobs of its neighbours)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // cut scan of src/refine.c:161-178: the first i in [lo, hi] with the smallest total, ties by the
    // larger Burset frequency.  The lanes take i = lo + lane, lo + lane + 64, ... (each reads its four
    // genomic characters at once instead of lane 0 walking <= len_p+1 dependent loads) and then agree.
    {
      const uint32_t avail = len_t + min(job.tail, 2u);
      const uint32_t lo = job.p0, hi = job.p1 > job.p0 ? job.p1 : job.p0;   // i = lo is always a candidate
      uint32_t bi = 0xFFFFFFFFu, bc = 0xFFFFFFFFu; int bf = -1;
      for (uint32_t i = lo + lane; i <= hi; i += 64) {
        const int freq = burset_adaptor(job.b, avail, pre_pos[i], len_t - suf_pos[len_p - i]);
        const uint32_t c = pre[i] + suf[len_p - i];
        if (bc > c || (bc == c && freq > bf)) { bc = c; bf = freq; bi = i; }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t oc = __shfl_xor(bc, off), oi = __shfl_xor(bi, off);
        const int of = __shfl_xor(bf, off);
        if (oc < bc || (oc == bc && (of > bf || (of == bf && oi < bi)))) { bc = oc; bf = of; bi = oi; }
      }
      if (lane == 0) {
        const uint32_t off_t1 = pre_pos[bi], off_t2 = suf_pos[len_p - bi];
        res->status = 0;
        res->v[0] = bc <= max_errs ? 1 : 0;
        res->v[1] = (int32_t)bi; res->v[2] = (int32_t)off_t1;
        res->v[3] = (int32_t)(len_t - off_t2); res->v[4] = (int32_t)bc;
      }
    }
  } else if constexpr (MODE == MODE_KBAND) {
    // K_band_edit_distance (src/compute-alignments.c:319-453): early exits in the reference's
    // order, then the banded DP (or the full matrix when 2k+1 >= n, :370-373).  rows = shorter.
    if (job.tail != 0u) {                                    // exon check: the two dust comparisons beside the distance
      const uint32_t fl = dust_flags_wave(job, lane);
      if (lane == 0) res->v[2] = (int32_t)fl;
    }
    const uint32_t ub = job.p0;
    const bool swap = job.la < job.lb;                       // reference: seq1 becomes the longer
    const uint8_t* lng = swap ? job.b : job.a; const uint8_t* sht = swap ? job.a : job.b;
    const uint32_t n = swap ? job.lb : job.la, m = swap ? job.la : job.lb;
    bool same = n == m;
    if (same) for (uint32_t i = lane; i < n; i += 64) same = same && lng[i] == sht[i];
    same = __all(same);
    if (same || ub == 0 || n - m > ub) {
      if (lane == 0) {
        res->status = 0;
        if (same) { res->v[0] = 1; res->v[1] = 0; }
        else if (ub == 0) { res->v[0] = 0; res->v[1] = 1; }
        else { res->v[0] = 0; res->v[1] = (int32_t)(n - m); }
      }
      return;
    }
    const bool banded = !(2ull * ub + 1 >= n);
    if (banded && 2u * ub + 1u <= 64u) { kband_band_sweep(lng, n, sht, m, ub, lane, res); return; }
    const Operand rows{sht, 0, false}, cols{lng, 0, false};
    if constexpr (STRIPS) {
      constexpr uint32_t SR = 64u * R;
      uint32_t* bnd = reinterpret_cast<uint32_t*>(ws + job.ws_off);
      const size_t bw = strip_bnd_bytes(n) / 4;
      uint32_t done = 0;
      for (uint32_t k = 0; done < m; ++k, done += SR) {
        const uint32_t part = min(SR, m - done);
        const Operand rs{sht + done, 0, false};
        const uint32_t* tp = k ? bnd + ((k - 1) & 1) * bw : nullptr;
        uint32_t* bt = done + part < m ? bnd + (k & 1) * bw : nullptr;
        if (banded) lev_sweep_strip<R, false, false, false, false, true>(rs, part, cols, n, lane, cur, minv, minpos, best, nullptr, ub, done, tp, bt);
        else        lev_sweep_strip<R, false, false, false, false, false>(rs, part, cols, n, lane, cur, minv, minpos, best, nullptr, 0, done, tp, bt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the next strip reads this strip's last row
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      const uint32_t lrow = m - (done - SR);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lane * R + r + 1 == lrow) { res->status = 0; res->v[1] = (int32_t)cur[r]; res->v[0] = cur[r] <= ub ? 1 : 0; }
      return;
    }
    if (banded) lev_sweep<R, false, false, false, false, true>(rows, m, cols, n, lane, cur, minv, minpos, best, nullptr, ub);
    else        lev_sweep<R, false, false, false, false, false>(rows, m, cols, n, lane, cur, minv, minpos, best, nullptr);
    if (m == 0) { if (lane == 0) { res->status = 0; res->v[1] = (int32_t)n; res->v[0] = n <= ub ? 1 : 0; } return; }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (lane * R + r + 1 == m) { res->status = 0; res->v[1] = (int32_t)cur[r]; res->v[0] = cur[r] <= ub ? 1 : 0; }
  } else {  // MODE_AFFIX
    const Operand rows{job.a, 0, false}, cols{job.b, 0, false};
    if constexpr (STRIPS) {
      constexpr uint32_t SR = 64u * R;
      uint32_t* bnd = reinterpret_cast<uint32_t*>(ws + job.ws_off);
      const size_t bw = strip_bnd_bytes(job.lb) / 4;
      uint32_t done = 0;
      for (uint32_t k = 0; done < job.la; ++k, done += SR) {
        const uint32_t part = min(SR, job.la - done);
        const Operand rs{job.a + done, 0, false};
        lev_sweep_strip<R, false, false, false, true>(rs, part, cols, job.lb, lane, cur, minv, minpos, best, nullptr, 0, done,
                                                k ? bnd + ((k - 1) & 1) * bw : nullptr, done + part < job.la ? bnd + (k & 1) * bw : nullptr);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the next strip reads this strip's last row
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    } else if (job.la + job.lb < 32768u) {
      lev_sweep<R, false, false, false, true, false, true>(rows, job.la, cols, job.lb, lane, cur, minv, minpos, best, nullptr);
    } else {
      lev_sweep<R, false, false, false, true>(rows, job.la, cols, job.lb, lane, cur, minv, minpos, best, nullptr);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {       // wave-wide arg-best
      const AffixBest o = best.from_lane_xor(off);
      if (o.s && best.worse_than(o)) best = o;
    }
    if (lane == 0) {
      res->status = 0; res->v[0] = (int32_t)best.valid();
      res->v[1] = (int32_t)best.e(); res->v[2] = (int32_t)best.g();
    }
  }
}

// one row class per launch (BORDERS with up to 64 rows, AFFIX with up to 64 rows or in strips)
template <int R, int MODE, bool STRIPS = false>
__global__ __launch_bounds__(MODE == MODE_BORDERS ? 64 : 256)
void lev_wave_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results,
                     uint8_t* __restrict__ ws) {
  constexpr int WAVES = MODE == MODE_BORDERS ? 1 : 4;
  const uint32_t lane = threadIdx.x & 63u;
  const int w = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (w >= njobs) return;
  const DevJob job = jobs[w];
  lev_wave_body<R, MODE, STRIPS>(job, &results[job.out_idx], ws, lane);
}

// ALL row classes of a family in ONE launch: the jobs of a merged batch are few per class, and
// kernels of one stream or hardware queue run one after the other, so a launch per class costs
// the sum of the classes' longest jobs; in one launch they overlap.  Every wave picks the body of
// its job's class (jobs are sorted by class, so the waves of a workgroup mostly agree).
constexpr int TB_WIN_BYTES = 8192;    // traceback: direction window per wave
// traceback: path steps buffered before the lanes write them out.  704, not 1024: with it a workgroup of dp_batch_kernel
// takes 39 744 + 256 B of LDS, and FOUR of them share a CU's 160 KB (16 job waves, what the registers allow) instead of three
constexpr int TB_PATH = 704;
__device__ __forceinline__ void align_traceback_wave(const DevJob& job, DevResult* res, const uint8_t* __restrict__ ws,
                                                     uint8_t* __restrict__ strs, const uint32_t lane,
                                                     uint8_t* win, uint8_t* path);
__device__ __forceinline__ void gap_traceback_wave(const DevJob& job, DevResult* res, const uint8_t* __restrict__ ws,
                                                   uint8_t* __restrict__ strs, const uint32_t lane,
                                                   uint8_t* win, uint8_t* path);

// The wave that filled a traceback workspace walks it right away (ALIGN, GAP): what it stored has
// to be visible to its other lanes, and lines of an earlier batch may sit in this CU's L1.
__device__ __forceinline__ void own_stores_visible() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// BIG = the classes with 32 and 64 rows per lane and the strips (rare in a batch, a few hundred
// registers per lane); the common classes up to 16 rows per lane get a kernel of their own whose
// register budget lets several waves share a SIMD.  Jobs are sorted big classes first, so the two
// launches take the two ends of the family's slice.
template <int MODE, bool BIG>
__device__ __forceinline__ void lev_any_dispatch(const DevJob& job, DevResult* res, uint8_t* __restrict__ ws, const uint32_t lane) {
  if constexpr (BIG) {
    switch (job.r_class) {
      case 32: lev_wave_body<32, MODE>(job, res, ws, lane); break;
      case 64: lev_wave_body<64, MODE>(job, res, ws, lane); break;
      default: lev_wave_body<64, MODE, true>(job, res, ws, lane); break;      // ROW_CLASS_STRIPS
    }
  } else {
    switch (job.r_class) {
      case 1:  lev_wave_body<1, MODE>(job, res, ws, lane); break;
      case 2:  lev_wave_body<2, MODE>(job, res, ws, lane); break;
      case 4:  lev_wave_body<4, MODE>(job, res, ws, lane); break;
      case 8:  lev_wave_body<8, MODE>(job, res, ws, lane); break;
      default: lev_wave_body<16, MODE>(job, res, ws, lane); break;
    }
  }
}

template <int MODE, bool BIG>
__global__ __launch_bounds__(256)
void lev_any_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results,
                    uint8_t* __restrict__ ws, uint8_t* __restrict__ strs) {
  const uint32_t lane = threadIdx.x & 63u;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= njobs) return;
  const DevJob job = jobs[w];
  DevResult* res = &results[job.out_idx];
  lev_any_dispatch<MODE, BIG>(job, res, ws, lane);
  if constexpr (MODE == MODE_ALIGN) {      // matrix, then the traceback by the same wave
    __shared__ __attribute__((aligned(16))) uint8_t s_win[4][TB_WIN_BYTES];
    __shared__ uint8_t s_path[4][TB_PATH];
    own_stores_visible();
    align_traceback_wave(job, res, ws, strs, lane, s_win[threadIdx.x >> 6], s_path[threadIdx.x >> 6]);
  }
}

// ---------------------------------------------------------------------------------------------
// Cooperative sweeps: one job over the W waves of a workgroup
// ---------------------------------------------------------------------------------------------
// A single wave needs (columns + 63) * R sequential cell updates for a job with 64*R rows; a merged
// batch holds only a few dozen jobs with hundreds of rows (small-exon searches, affix recovery),
// so those launches are latency-bound and sit on the critical path of every batch.  Here the
// strip of rows is spread over W*64 lanes: wave w owns lanes 64w .. 64w+63 of the same skewed
// sweep.  Lane 0 of wave w+1 needs, at step s, what lane 63 of wave w produced at step s-1; the
// waves are decoupled by one 64-step chunk: in interval k wave w works on chunk k-w and leaves the
// packed (value | column char) of its last lane, per step, in an LDS array that wave w+1 reads
// in the next interval (double-buffered; one __syncthreads per interval).  The column characters
// travel with the wavefront, so only wave 0 reads them from memory.
constexpr int COOP_W = 4;

template <int R, bool ROWMIN, bool AFFIX, bool ASMALL = false, bool WILD = false, bool DIRS = false>
__device__ __forceinline__ void lev_sweep_coop(const Operand rows, const uint32_t nr,
                                               const Operand cols, const uint32_t nc,
                                               const uint32_t w, const uint32_t lane,
                                               uint32_t* hand,        // [COOP_W-1][2][64] of this sweep
                                               uint32_t (&cur)[R], uint32_t (&minv)[R],
                                               uint32_t (&minpos)[R], AffixBest& best,
                                               uint8_t* dir_ws = nullptr) {   // DIRS: [step][256 lanes] entries
  uint32_t rc[R];
  const uint32_t gl = w * 64u + lane;   // lane index within the job (w is wave-uniform)
  const uint32_t row0 = gl * R;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = row0 + r;
    rc[r] = i < nr ? rows.at(i) : PAD_ROW;
    cur[r] = i + 1;
    if constexpr (ROWMIN) { minv[r] = i + 1; minpos[r] = 0; }
  }
  if (nr == 0 || nc == 0) return;       // uniform over the workgroup (one job)
  const uint32_t last_lane = (nr - 1) / R;
  const uint32_t steps = nc + last_lane;
  const uint32_t nchunks = (steps + 63u) / 64u;
  const bool wave_used = w * 64u <= last_lane;
  uint32_t diag_in = row0, out = 0;
  // The step loop is a chain of dependent instructions on one wave, so every instruction in it is
  // latency, and the hand-off between the waves costs ONE DPP move per step on either side:
  //  * producer: `hout` is a shift register, hout = (lanes move one down, lane 63 <- this step's
  //    `out` of lane 63); after the T steps of a chunk lane 63-k holds the step k before the last.
  //    It goes to the LDS array once per chunk (all lanes, one store);
  //  * consumer: reads the array once per chunk and turns it (one ds_bpermute) into `feed`, lane t =
  //    the input of step t = the producer's output of the step before; `feed` moves one lane down
  //    per step so lane 0 always holds the current input, and the DPP shift that passes the strips'
  //    last rows along drops it in.  Wave 0's feed is (top value | column character) instead.
  // No per-step LDS access, scalar round trip or branch on the wave number.  A wave's first cell
  // lies in chunk w; it looks at the array one chunk earlier for the step before its first.
  uint32_t hout = 0, prev = 0;
  uint32_t feed_next = (lane + 1u) | ((lane < nc ? cols.at(lane) : PAD_COL) << 24);   // wave 0: columns of chunk 0
  uint32_t* hand_in = hand + (w > 0 ? (w - 1) * 128u : 0u);
  uint32_t* hand_out = hand + (w + 1 < (uint32_t)COOP_W ? w : 0u) * 128u;

  for (uint32_t k = 0; k < nchunks + COOP_W - 1; ++k) {
    const int c = (int)k - (int)w;
    if (wave_used && w > 0 && c == (int)w - 1) prev = hand_in[(c & 1) * 64 + lane];
    if (wave_used && c >= (int)w && c < (int)nchunks) {
      const uint32_t s0 = (uint32_t)c * 64u;
      const uint32_t tmax = (min(64u, steps - s0) + 7u) & ~7u;   // whole groups of 8; steps past the end touch no cell
      uint32_t feed;
      if (w == 0) {
        feed = feed_next;                                    // requested one chunk ago (see lev_sweep)
        const uint32_t jn = s0 + 64u + lane;
        feed_next = (jn + 1u) | ((jn < nc ? cols.at(jn) : PAD_COL) << 24);
      } else {
        // the producer ran the same tmax steps on this chunk: its step t is in lane 64-tmax+t, the
        // last step of the chunk before in lane 63-tmax -- or, after a full chunk, in lane 63 of
        // what was read for the chunk before
        const uint32_t cur_in = hand_in[(c & 1) * 64 + lane];
        feed = (uint32_t)__shfl((int)cur_in, (int)((lane + 63u - tmax) & 63u));
        const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)prev, 63);
        if (tmax == 64u && lane == 0) feed = carry;
        prev = cur_in;
      }
      for (uint32_t t0 = 0; t0 < tmax; t0 += 8) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
          const uint32_t s = s0 + t0 + u;
          const uint32_t in = wave_shr1_first(feed, out);
          feed = wave_shl1(feed);
          const uint32_t j = s - gl + 1;
          if (j - 1u < nc) {
            const uint32_t ch = in >> 24;
            const uint32_t in_val = in & 0xFFFFFFu;
            uint32_t up = in_val;
            uint32_t diag = diag_in;
            const bool ch_n = WILD && is_n(ch);
            DirPack<R> dp;
            if constexpr (DIRS) dp.clear();
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const uint32_t left = cur[r];
              bool match = rc[r] == ch;
              if constexpr (WILD) match = match || ch_n || is_n(rc[r]);
              uint32_t v = diag + (match ? 0u : 1u);
              if constexpr (DIRS) {
                // ComputeAlignMatrix tie-break: diagonal, then up (dir 1), then left (dir 2), strict >
                uint32_t d = 0;
                if (v > up + 1) { v = up + 1; d = 1; }
                if (v > left + 1) { v = left + 1; d = 2; }
                dp.set(r, d);
              } else {
                v = min(v, min(up + 1, left + 1));
              }
              if constexpr (ROWMIN) {
                if (minv[r] > v) { minv[r] = v; minpos[r] = j; }
              }
              if constexpr (AFFIX) best.template consider<ASMALL>(rc[r] == ch, v, row0 + r + 1, j);
              diag = left;
              cur[r] = v;
              up = v;
            }
            diag_in = in_val;
            out = up | (ch << 24);
            if constexpr (DIRS) dp.store(dir_ws + ((size_t)s * (64u * COOP_W) + gl) * (R <= 4 ? 1u : R / 4));
          }
          hout = wave_shl1_last(out, hout);
        }
      }
      if (w + 1 < (uint32_t)COOP_W) hand_out[(c & 1) * 64 + lane] = hout;
    }
    __syncthreads();
  }
}

// general_refine_borders (src/refine.c:105-192) for patterns longer than 64: the prefix sweep on
// waves 0-3 and the reversed-string sweep on waves 4-7 of one 512-thread workgroup.
template <int R>
__device__ __forceinline__ void borders_coop_body(const DevJob& job, DevResult* res) {
  extern __shared__ uint32_t lds[];      // hand-off [2][COOP_W-1][2][64], then pre, pre_pos, suf, suf_pos
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: branches on it are uniform
  const uint32_t sweep = wave / COOP_W, w = wave % COOP_W;
  const uint32_t len_p = job.la, len_t = job.lb, max_errs = job.p2;
  const uint32_t t_win = min(len_p + max_errs, len_t);
  uint32_t* hand = lds + sweep * ((COOP_W - 1) * 128);
  uint32_t* pre = lds + 2 * (COOP_W - 1) * 128; uint32_t* pre_pos = pre + (len_p + 1);
  uint32_t* suf = pre_pos + (len_p + 1); uint32_t* suf_pos = suf + (len_p + 1);
  uint32_t cur[R], minv[R], minpos[R];
  AffixBest best = AFFIX_NONE;
  const Operand rows{job.a, len_p, sweep == 1}, cols{job.b, len_t, sweep == 1};
  lev_sweep_coop<R, true, false>(rows, len_p, cols, t_win, w, lane, hand, cur, minv, minpos, best);
  uint32_t* mv = sweep ? suf : pre; uint32_t* mp = sweep ? suf_pos : pre_pos;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = (w * 64u + lane) * R + r + 1;
    if (i <= len_p) { mv[i] = minv[r]; mp[i] = minpos[r]; }
  }
  if (threadIdx.x == 0) { pre[0] = 0; pre_pos[0] = 0; suf[0] = 0; suf_pos[0] = 0; }
  __syncthreads();
  if (wave != 0) return;
  // cut scan of src/refine.c:161-178: the first i in [lo, hi] with the smallest total, ties by the
  // larger Burset frequency; 64 lanes scan i = lo + lane, lo + lane + 64, ... and then agree
  const uint32_t avail = len_t + min(job.tail, 2u);
  const uint32_t lo = job.p0, hi = job.p1 > job.p0 ? job.p1 : job.p0;   // i = lo is always a candidate
  uint32_t bi = 0xFFFFFFFFu, bc = 0xFFFFFFFFu; int bf = -1;
  for (uint32_t i = lo + lane; i <= hi; i += 64) {
    const int freq = burset_adaptor(job.b, avail, pre_pos[i], len_t - suf_pos[len_p - i]);
    const uint32_t c = pre[i] + suf[len_p - i];
    if (bc > c || (bc == c && freq > bf)) { bc = c; bf = freq; bi = i; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t oc = __shfl_xor(bc, off), oi = __shfl_xor(bi, off);
    const int of = __shfl_xor(bf, off);
    if (oc < bc || (oc == bc && (of > bf || (of == bf && oi < bi)))) { bc = oc; bf = of; bi = oi; }
  }
  if (lane == 0) {
    const uint32_t off_t1 = pre_pos[bi], off_t2 = suf_pos[len_p - bi];
    res->status = 0;
    res->v[0] = bc <= max_errs ? 1 : 0;
    res->v[1] = (int32_t)bi; res->v[2] = (int32_t)off_t1;
    res->v[3] = (int32_t)(len_t - off_t2); res->v[4] = (int32_t)bc;
  }
}

// every row class above 64 rows in one launch (one job per workgroup; see lev_any_kernel)
__device__ __forceinline__ void borders_coop_dispatch(const DevJob& job, DevResult* res) {
  switch (job.r_class) {                 // rows per lane of the 256-lane sweep = class / COOP_W
    case 2: case 4: borders_coop_body<1>(job, res); break;
    case 8:  borders_coop_body<2>(job, res); break;
    case 16: borders_coop_body<4>(job, res); break;
    case 32: borders_coop_body<8>(job, res); break;
    default: borders_coop_body<16>(job, res); break;
  }
}

__global__ __launch_bounds__(512)
void borders_coop_any_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results) {
  const DevJob job = jobs[blockIdx.x];
  borders_coop_dispatch(job, &results[job.out_idx]);
}

// find_longest_affix (src/factorization-refinement.c:1136-1173) for more than 64 rows
template <int R>
__device__ __forceinline__ void affix_coop_body(const DevJob& job, DevResult* res, uint32_t* hand, uint32_t (*wbest)[5]) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t cur[R], minv[R], minpos[R];
  AffixBest best = AFFIX_NONE;
  const Operand rows{job.a, 0, false}, cols{job.b, 0, false};
  // e + g < 2^15: every product of the cut test fits 24 x 24 -> 32 bits (AffixBest::consider)
  if (job.la + job.lb < 32768u) lev_sweep_coop<R, false, true, true>(rows, job.la, cols, job.lb, w, lane, hand, cur, minv, minpos, best);
  else                          lev_sweep_coop<R, false, true, false>(rows, job.la, cols, job.lb, w, lane, hand, cur, minv, minpos, best);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const AffixBest o = best.from_lane_xor(off);
    if (o.s && best.worse_than(o)) best = o;
  }
  if (lane == 0) { wbest[w][0] = best.v; wbest[w][1] = best.s; wbest[w][2] = best.e(); wbest[w][3] = best.g(); }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < COOP_W; ++k) {
      const AffixBest o{wbest[k][0], wbest[k][1], ((uint64_t)wbest[k][2] << 32) | wbest[k][3]};
      if (o.s && best.worse_than(o)) best = o;
    }
    res->status = 0; res->v[0] = (int32_t)best.valid();
    res->v[1] = (int32_t)best.e(); res->v[2] = (int32_t)best.g();
  }
}

__device__ __forceinline__ void affix_coop_dispatch(const DevJob& job, DevResult* res, uint32_t* hand, uint32_t (*wbest)[5]) {
  switch (job.r_class) {
    case 2: case 4: affix_coop_body<1>(job, res, hand, wbest); break;
    case 8:  affix_coop_body<2>(job, res, hand, wbest); break;
    case 16: affix_coop_body<4>(job, res, hand, wbest); break;
    case 32: affix_coop_body<8>(job, res, hand, wbest); break;
    default: affix_coop_body<16>(job, res, hand, wbest); break;
  }
}

__global__ __launch_bounds__(256)
void affix_coop_any_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results) {
  __shared__ uint32_t hand[(COOP_W - 1) * 128];
  __shared__ uint32_t wbest[COOP_W][5];
  const DevJob job = jobs[blockIdx.x];
  affix_coop_dispatch(job, &results[job.out_idx], hand, wbest);
}

// TracebackAlignment (src/compute-alignments.c:149-207), one WAVE per job.
// The walk from (n,m) back to the border is a chain of dependent direction look-ups; done by one
// thread against HBM/L2 every step costs a memory round trip (~250 ns).  Here the wave copies a
// window of direction entries (a run of consecutive sweep steps, coalesced 16 B per lane) into
// LDS and walks it there; the walk state is wave-uniform, so it lives in scalar registers and a
// step is one LDS read plus a few scalar instructions.  The walk only records the 2-bit
// direction per step; the gapped strings are then written by all 64 lanes at once: the character
// a step consumes is found from a prefix count (ballot + popcount) of the steps before it.

__device__ __forceinline__ void tb_flush(const uint8_t* path, uint32_t np, const uint8_t* a, const uint8_t* b,
                                         uint32_t i0, uint32_t j0, uint32_t pos0, uint8_t* ea, uint8_t* ga,
                                         uint32_t lane) {
  uint32_t ca = 0, cb = 0;             // characters of a / b consumed by the steps before this group
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t base = 0; base < np; base += 64) {
    const uint32_t idx = base + lane;
    const bool valid = idx < np;
    const uint32_t d = valid ? path[idx] : 1u;
    const bool ua = valid && d <= 1u, ub = valid && d != 1u;       // step consumes a[..] / b[..] (2, 3: b only)
    const unsigned long long ma = __ballot(ua), mb = __ballot(ub);
    if (valid) {
      const uint32_t ia = i0 - 1u - (ca + (uint32_t)__popcll(ma & lt));
      const uint32_t ib = j0 - 1u - (cb + (uint32_t)__popcll(mb & lt));
      ea[pos0 - 1u - idx] = ua ? a[ia] : (uint8_t)'-';
      ga[pos0 - 1u - idx] = ub ? b[ib] : (uint8_t)'-';
    }
    ca += (uint32_t)__popcll(ma); cb += (uint32_t)__popcll(mb);
  }
}

__device__ __forceinline__ void align_traceback_wave(const DevJob& job, DevResult* res, const uint8_t* __restrict__ ws,
                                                     uint8_t* __restrict__ strs, const uint32_t lane,
                                                     uint8_t* win, uint8_t* path) {
  // the walk is wave-uniform: its state has to be uniform for the compiler too (scalar registers and
  // branches instead of per-lane values under an exec mask: a third of the instructions per step)
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.la), m = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.lb), cap = n + m + 1;
  uint8_t* ea = strs + job.str_off;
  uint8_t* ga = ea + cap;
  if (res->v[5] == 1) {                  // identity alignment
    for (uint32_t i = lane; i < n; i += 64) { ea[i] = job.a[i]; ga[i] = job.b[i]; }
    if (lane == 0) {
      ea[n] = 0; ga[n] = 0;
      res->v[1] = (int32_t)n;
      res->str[0] = job.str_off; res->str[1] = job.str_off + cap;
      res->v[5] = 0;
    }
    return;
  }
  const uint32_t rcls = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.r_class);
  const bool strips = rcls == ROW_CLASS_STRIPS;            // more than 4096 rows: strips of 64*64 rows
  const uint32_t R = strips ? 64u : rcls, EB = R <= 4 ? 1u : R / 4;
  const uint32_t lgR = 31u - (uint32_t)__builtin_clz(R);   // R is a power of two
  const uint32_t WS = TB_WIN_BYTES / (64u * EB);           // sweep steps per window
  const uint8_t* dirs = ws + job.ws_off + (strips ? 2 * strip_bnd_bytes(m) : 0);
  const size_t strip_dirs = ((size_t)m + 64) * 64 * EB;
  uint32_t i = n, j = m, k = 0, np = 0;
  uint32_t i0 = n, j0 = m, pos = cap - 1;
  if (lane == 0) { ea[pos] = 0; ga[pos] = 0; }
  uint32_t s_lo = 1u, s_hi = 0u, win_strip = 0;            // empty window
  while (i > 0 && j > 0) {
    const uint32_t strip = strips ? (i - 1) >> 12 : 0u, li = strips ? (i - 1) & 4095u : i - 1;
    const uint32_t l = li >> lgR, r = li & (R - 1), s = (j - 1) + l;
    if (s < s_lo || s > s_hi || strip != win_strip) {      // bring in the steps (s - WS, s] of the strip
      s_hi = s; s_lo = s + 1 >= WS ? s + 1 - WS : 0; win_strip = strip;
      const uint32_t bytes = (s_hi - s_lo + 1) * 64u * EB;
      const uint8_t* src = dirs + strip * strip_dirs + (size_t)s_lo * 64u * EB;
      for (uint32_t off = lane * 16u; off < bytes; off += 64u * 16u)
        *reinterpret_cast<uint4*>(win + off) = *reinterpret_cast<const uint4*>(src + off);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t d = (win[((s - s_lo) * 64u + l) * EB + (r >> 2)] >> (2u * (r & 3u))) & 3u;
    d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);  // the walk is wave-uniform: keep it scalar
    path[np] = (uint8_t)d;                                // every lane, same address, same value: no exec games
    ++np;
    i -= d < 2u ? 1u : 0u;                                 // 0: diagonal, 1: up, 2: left -- no branches in the step
    j -= d != 1u ? 1u : 0u;
    if (np == TB_PATH) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
      pos -= np; k += np; np = 0; i0 = i; j0 = j;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
  pos -= np; k += np;
  // what is left of one string is aligned to gaps (:193-206): a[..i) over '-', then '-' over b[..j)
  for (uint32_t q = lane; q < i; q += 64) { ea[pos - 1 - q] = job.a[i - 1 - q]; ga[pos - 1 - q] = '-'; }
  pos -= i; k += i;
  for (uint32_t q = lane; q < j; q += 64) { ea[pos - 1 - q] = '-'; ga[pos - 1 - q] = job.b[j - 1 - q]; }
  pos -= j; k += j;
  if (lane == 0) {
    res->v[1] = (int32_t)k;
    res->str[0] = job.str_off + pos;
    res->str[1] = job.str_off + cap + pos;
  }
}

// ---------------------------------------------------------------------------------------------
// ComputeAlignMatrix + TracebackAlignment (src/compute-alignments.c:85-207) for more than 64 rows: the sweep
// on the four waves of a workgroup (lev_sweep_coop with the N wildcard and the direction stream), the
// traceback by wave 0.  One wave needs rows/64 cells per step in a dependent chain (a 250-row exon: four
// cells per lane and step, 125 us); 256 lanes take one.  Directions are stored step-major over 256 lanes,
// [step][lane], 2 bits per row of the lane; the walk only ever needs the lanes at and below its own within a
// run of steps, so it stages a band of 64 lanes x (8192 / (64 x entry bytes)) steps in LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t align_coop_rows_per_lane(uint32_t r_class) { return r_class <= 4u ? 1u : r_class / 4u; }

__device__ __forceinline__ void align_traceback_coop(const DevJob& job, DevResult* res, const uint8_t* __restrict__ ws,
                                                     uint8_t* __restrict__ strs, const uint32_t lane,
                                                     uint8_t* win, uint8_t* path) {
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.la), m = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.lb), cap = n + m + 1;
  uint8_t* ea = strs + job.str_off;
  uint8_t* ga = ea + cap;
  const uint32_t R = align_coop_rows_per_lane((uint32_t)__builtin_amdgcn_readfirstlane((int)job.r_class)), EB = R <= 4 ? 1u : R / 4;
  const uint32_t lgR = 31u - (uint32_t)__builtin_clz(R);
  const uint32_t WS = TB_WIN_BYTES / (64u * EB);           // sweep steps per window (band of 64 lanes)
  const uint8_t* dirs = ws + job.ws_off;
  const uint32_t LT = 64u * COOP_W;                        // lanes of the sweep
  uint32_t i = n, j = m, k = 0, np = 0;
  uint32_t i0 = n, j0 = m, pos = cap - 1;
  if (lane == 0) { ea[pos] = 0; ga[pos] = 0; }
  uint32_t s_lo = 1u, s_hi = 0u, g_lo = 0u;                // empty window
  while (i > 0 && j > 0) {
    const uint32_t gl = (i - 1) >> lgR, r = (i - 1) & (R - 1), s = (j - 1) + gl;
    if (s < s_lo || s > s_hi || gl < g_lo || gl >= g_lo + 64u) {
      // steps (s - WS, s], lanes [g_lo, g_lo + 64) with g_lo a multiple of 16 at least 32 below gl
      s_hi = s; s_lo = s + 1 >= WS ? s + 1 - WS : 0;
      g_lo = (gl & ~15u) >= 48u ? (gl & ~15u) - 48u : 0u;
      const uint32_t row16 = 4u * EB;                      // 16-byte pieces per step row of the band
      const uint32_t total16 = (s_hi - s_lo + 1) * row16;
      for (uint32_t e = lane; e < total16; e += 64u) {
        const uint32_t row = e / row16, c16 = e % row16;
        *reinterpret_cast<uint4*>(win + (size_t)e * 16u) =
            *reinterpret_cast<const uint4*>(dirs + ((size_t)(s_lo + row) * LT + g_lo) * EB + c16 * 16u);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t d = (win[((s - s_lo) * 64u + (gl - g_lo)) * EB + (r >> 2)] >> (2u * (r & 3u))) & 3u;
    d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
    path[np] = (uint8_t)d;                                // every lane, same address, same value: no exec games
    ++np;
    i -= d < 2u ? 1u : 0u;                                 // 0: diagonal, 1: up, 2: left -- no branches in the step
    j -= d != 1u ? 1u : 0u;
    if (np == TB_PATH) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
      pos -= np; k += np; np = 0; i0 = i; j0 = j;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
  pos -= np; k += np;
  for (uint32_t q = lane; q < i; q += 64) { ea[pos - 1 - q] = job.a[i - 1 - q]; ga[pos - 1 - q] = '-'; }
  pos -= i; k += i;
  for (uint32_t q = lane; q < j; q += 64) { ea[pos - 1 - q] = '-'; ga[pos - 1 - q] = job.b[j - 1 - q]; }
  pos -= j; k += j;
  if (lane == 0) {
    res->v[1] = (int32_t)k;
    res->str[0] = job.str_off + pos;
    res->str[1] = job.str_off + cap + pos;
  }
}

// ---------------------------------------------------------------------------------------------
// ComputeAlignMatrix + TracebackAlignment (src/compute-alignments.c:85-207) inside a BAND, on one wave.
// An exon and its stretch of the genomic sequence differ by a few per cent, so the alignment lies next to
// the main diagonal.  Lane s owns the diagonal column - row = s - k (k = ALIGN_BAND_K, 2k+1 <= 64) and the
// wave walks down the rows exactly as kband_band_sweep does (lane s takes row r at time 2r + s; diagonal =
// its own previous value, up = the right neighbour's previous row, left = the left neighbour's same row),
// with the N wildcard and the reference's preference diagonal < up < left recorded per cell.
// Why the answer is the full matrix's: M[i][j] >= |i - j|, so a cell whose true value is <= k lies inside
// the band together with every optimal path that ends in it -- its banded value is the true one; a cell
// with a true value > k gets a banded value >= the true one, hence > k.  If the banded M[n][m] is <= k it
// is the true score, every cell of the reference's traceback has a value <= the score, and at such a cell
// the candidates that reach the minimum have true (= banded) neighbour values, all others are larger in
// both matrices (a neighbour outside the band is worth >= k + 1): the same first minimum in the order
// diagonal, up, left, i.e. the same direction.  A banded score > k says nothing: the caller sweeps the
// whole matrix.  Directions: 2 bits per cell, one 32-bit word per lane and 16 rows, [row / 16][lane].
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ALIGN_BAND_K = ALIGN_BAND_HALF;
__device__ __forceinline__ bool align_band_fits(uint32_t n, uint32_t m) {
  return n > 0u && m > 0u && (n > m ? n - m : m - n) <= ALIGN_BAND_K;
}

// returns the banded M[n][m] in every lane; dirs: ((n >> 4) + 1) * 64 words
__device__ __noinline__ uint32_t align_band_sweep(const uint8_t* __restrict__ a, const uint32_t n,
                                                  const uint8_t* __restrict__ b, const uint32_t m,
                                                  const uint32_t lane, uint32_t* __restrict__ dirs) {
  constexpr uint32_t k = ALIGN_BAND_K, W = 2u * k + 1u;
  const bool used = lane < W;
  const bool odd = (lane & 1u) != 0u;
  const int off = (int)lane - (int)k;            // column - row on this lane's diagonal
  const uint32_t half = lane >> 1;               // iteration q works on row r = q - half
  const uint32_t up_mask = (lane + 1u < W) ? 0u : BAND_INF, left_mask = lane > 0u ? 0u : BAND_INF;
  uint32_t val = (uint32_t)(off < 0 ? -off : off);
  // rows of this lane: 1 <= r <= n with 1 <= r + off <= m
  const uint32_t r_lo = off < 0 ? (uint32_t)(1 - off) : 1u;
  const int hi_i = (int)m - off < (int)n ? (int)m - off : (int)n;
  const bool any_row = used && hi_i >= (int)r_lo;
  const uint32_t span = any_row ? (uint32_t)hi_i - r_lo : 0xFFFFFFFFu;   // r - r_lo <= span: active
  const uint32_t nq = n + k;                     // lane 2k takes row n in iteration n + k
  auto row_char = [&](uint32_t q) -> uint32_t {
    const uint32_t r = q - half;
    return (any_row && r - r_lo <= span) ? (uint32_t)a[r - 1u] : 0u;
  };
  auto col_char = [&](uint32_t q) -> uint32_t {
    const uint32_t r = q - half;
    return (any_row && r - r_lo <= span) ? (uint32_t)b[(uint32_t)(off + (int)r) - 1u] : 1u;
  };
  uint32_t dw = 0u;                              // directions of the rows of the current group of 16
  uint32_t ca[4], cb[4], can[4], cbn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { ca[j] = row_char(1u + j); cb[j] = col_char(1u + j); }
  for (uint32_t q0 = 1; q0 <= nq; q0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { can[j] = row_char(q0 + 4u + j); cbn[j] = col_char(q0 + 4u + j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t r = q0 + j - half;          // wraps for the lanes whose first row lies ahead
      const bool active = any_row && (r - r_lo <= span);
      const uint32_t mism = (ca[j] == cb[j] || is_n(ca[j]) || is_n(cb[j])) ? 0u : 1u;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const uint32_t upv = (wave_shl1(val) | up_mask) + 1u, leftv = (wave_shr1(val) | left_mask) + 1u;
        uint32_t nv = val + mism, d = 0u;
        if (nv > upv) { nv = upv; d = 1u; }
        if (nv > leftv) { nv = leftv; d = 2u; }
        const bool mine = active && odd == (par == 1);
        val = mine ? nv : val;
        dw |= mine ? d << (2u * (r & 15u)) : 0u;
      }
      // the word of rows 16 g .. 16 g + 15 is complete after row 16 g + 15, or after the lane's last row
      if (active && ((r & 15u) == 15u || r - r_lo == span)) { dirs[(r >> 4) * 64u + lane] = dw; dw = 0u; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { ca[j] = can[j]; cb[j] = cbn[j]; }
  }
  return (uint32_t)__shfl((int)val, (int)(m + k - n));
}

// the traceback over the band's directions (same walk, same output as align_traceback_wave)
__device__ __forceinline__ void align_band_traceback(const DevJob& job, DevResult* res, const uint32_t* __restrict__ dirs,
                                                     uint8_t* __restrict__ strs, const uint32_t lane,
                                                     uint8_t* win, uint8_t* path) {
  constexpr uint32_t k0 = ALIGN_BAND_K;
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.la), m = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.lb), cap = n + m + 1;
  uint8_t* ea = strs + job.str_off;
  uint8_t* ga = ea + cap;
  constexpr uint32_t WB = TB_WIN_BYTES / 256u;             // groups of 16 rows per window
  const uint32_t* w32 = reinterpret_cast<const uint32_t*>(win);
  uint32_t i = n, j = m, k = 0, np = 0;
  uint32_t i0 = n, j0 = m, pos = cap - 1;
  if (lane == 0) { ea[pos] = 0; ga[pos] = 0; }
  uint32_t g_lo = 1u, g_hi = 0u;                           // empty window (groups of 16 rows)
  while (i > 0 && j > 0) {
    const uint32_t g = i >> 4;
    if (g < g_lo || g > g_hi) {                            // bring in the groups (g - WB, g]
      g_hi = g; g_lo = g + 1 >= WB ? g + 1 - WB : 0;
      const uint32_t bytes = (g_hi - g_lo + 1) * 256u;
      const uint8_t* src = reinterpret_cast<const uint8_t*>(dirs) + (size_t)g_lo * 256u;
      for (uint32_t off = lane * 16u; off < bytes; off += 64u * 16u)
        *reinterpret_cast<uint4*>(win + off) = *reinterpret_cast<const uint4*>(src + off);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t d = (w32[(g - g_lo) * 64u + (j + k0 - i)] >> (2u * (i & 15u))) & 3u;
    d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);  // the walk is wave-uniform: keep it scalar
    path[np] = (uint8_t)d;
    ++np;
    i -= d < 2u ? 1u : 0u;                                 // 0: diagonal, 1: up, 2: left
    j -= d != 1u ? 1u : 0u;
    if (np == TB_PATH) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
      pos -= np; k += np; np = 0; i0 = i; j0 = j;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
  pos -= np; k += np;
  for (uint32_t q = lane; q < i; q += 64) { ea[pos - 1 - q] = job.a[i - 1 - q]; ga[pos - 1 - q] = '-'; }
  pos -= i; k += i;
  for (uint32_t q = lane; q < j; q += 64) { ea[pos - 1 - q] = '-'; ga[pos - 1 - q] = job.b[j - 1 - q]; }
  pos -= j; k += j;
  if (lane == 0) {
    res->v[1] = (int32_t)k;
    res->str[0] = job.str_off + pos;
    res->str[1] = job.str_off + cap + pos;
  }
}

// one job on the first four waves of a workgroup; smem: hand-off (3 x 128 words), then the traceback's
// window (TB_WIN_BYTES, 16-aligned) and path (TB_PATH)
constexpr size_t ALIGN_COOP_LDS = (COOP_W - 1) * 128 * sizeof(uint32_t) + TB_WIN_BYTES + TB_PATH;
template <int R>
__device__ __forceinline__ void align_coop_body(const DevJob& job, DevResult* res, uint8_t* __restrict__ ws,
                                                uint8_t* __restrict__ strs, uint8_t* smem) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t* hand = reinterpret_cast<uint32_t*>(smem);
  uint8_t* win = smem + (COOP_W - 1) * 128 * sizeof(uint32_t);
  uint8_t* path = win + TB_WIN_BYTES;
  const uint32_t n = job.la, m = job.lb;
  // identity alignment, score 0 (compute-alignments.c:48-58)
  bool same = n == m;
  if (same) for (uint32_t q = threadIdx.x; q < n; q += 64u * COOP_W) same = same && job.a[q] == job.b[q];
  if (__syncthreads_and(same ? 1 : 0)) {
    if (w != 0) return;
    if (lane == 0) { res->status = 0; res->v[0] = 0; res->v[1] = (int32_t)n; res->v[5] = 1; }
    own_stores_visible();
    align_traceback_wave(job, res, ws, strs, lane, win, path);      // its identity branch
    return;
  }
  // first inside a band on wave 0 (job.tail: the host's switch); the other waves wait for its verdict
  if (job.tail != 0u && align_band_fits(n, m)) {
    uint32_t* bdirs = reinterpret_cast<uint32_t*>(ws + job.ws_off);
    if (w == 0) {
      const uint32_t score = align_band_sweep(job.a, n, job.b, m, lane, bdirs);
      if (lane == 0) hand[0] = score;
    }
    __syncthreads();
    const uint32_t score = hand[0];
    __syncthreads();                     // (hand is the sweep's hand-off buffer below)
    if (score <= ALIGN_BAND_K) {
      if (w != 0) return;
      if (lane == 0) { res->status = 0; res->v[0] = (int32_t)score; res->v[5] = 0; }
      own_stores_visible();
      align_band_traceback(job, res, bdirs, strs, lane, win, path);
      return;
    }
  }
  uint32_t cur[R], minv[R], minpos[R];
  AffixBest best = AFFIX_NONE;
  const Operand rows{job.a, 0, false}, cols{job.b, 0, false};
  lev_sweep_coop<R, false, false, false, true, true>(rows, n, cols, m, w, lane, hand, cur, minv, minpos, best, ws + job.ws_off);
  if (n == 0 || m == 0) { if (threadIdx.x == 0) { res->status = 0; res->v[0] = (int32_t)(n + m); res->v[5] = 0; } }
  else {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((w * 64u + lane) * R + r + 1 == n) { res->status = 0; res->v[0] = (int32_t)cur[r]; res->v[5] = 0; }
  }
  // the four waves' directions (and the score) have to be visible to wave 0
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (w != 0) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  align_traceback_coop(job, res, ws, strs, lane, win, path);
}

__device__ __forceinline__ void align_coop_dispatch(const DevJob& job, DevResult* res, uint8_t* __restrict__ ws,
                                                    uint8_t* __restrict__ strs, uint8_t* smem) {
  switch (job.r_class) {                 // rows per lane of the 256-lane sweep = class / COOP_W
    case 2: case 4: align_coop_body<1>(job, res, ws, strs, smem); break;
    case 8:  align_coop_body<2>(job, res, ws, strs, smem); break;
    case 16: align_coop_body<4>(job, res, ws, strs, smem); break;
    case 32: align_coop_body<8>(job, res, ws, strs, smem); break;
    default: align_coop_body<16>(job, res, ws, strs, smem); break;
  }
}

__global__ __launch_bounds__(256)
void align_coop_any_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results,
                           uint8_t* __restrict__ ws, uint8_t* __restrict__ strs) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[ALIGN_COOP_LDS];
  const DevJob job = jobs[blockIdx.x];
  align_coop_dispatch(job, &results[job.out_idx], ws, strs, smem);
}

// ---------------------------------------------------------------------------------------------
// 3-state gap alignment: ComputeGapAlignMatrix with only_one_align (src/refine-intron.c:623-824)
// ---------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void gap_wave_body(const DevJob& job, DevResult* res, uint8_t* __restrict__ ws,
                                              const uint32_t lane) {
  const uint32_t n = job.la, m = job.lb;
  int32_t cL[R], cG[R], cR[R];
  uint32_t rc[R];
  const uint32_t row0 = lane * R;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    rc[r] = row0 + r < n ? job.a[row0 + r] : PAD_ROW;
    cL[r] = 0; cG[r] = 0; cR[r] = 0;                       // column 0 of every plane is 0
  }
  if (n > 0 && m > 0) {
    const uint32_t last_lane = (n - 1) / R, steps = m + last_lane;
    int32_t dgL = 0, dgR = 0;                               // row above the strip, previous column
    uint32_t out = 0, outc = 0;
    uint8_t* dirs = ws + job.ws_off;
    // per 64-step chunk the column characters sit in `feed` (lane t = step t), which moves one lane
    // down per step: lane 0 always holds the current one (see lev_sweep)
    uint32_t feed_next = lane < m ? job.b[lane] : PAD_COL;        // requested one chunk ahead (see lev_sweep)
    for (uint32_t s0 = 0; s0 < steps; s0 += 64) {
      uint32_t feed = feed_next;
      const uint32_t jn = s0 + 64u + lane;
      feed_next = jn < m ? job.b[jn] : PAD_COL;
      const uint32_t tmax = (min(64u, steps - s0) + 7u) & ~7u;   // whole groups of 8; steps past the end touch no cell
      for (uint32_t t0 = 0; t0 < tmax; t0 += 8) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
          const uint32_t s = s0 + t0 + u;
          const uint32_t in = wave_shr1_first(0u, out);         // row 0 of L and R is 0
          const uint32_t inc = wave_shr1_first(feed, outc);
          feed = wave_shl1(feed);
          const uint32_t j = s - lane + 1;
          if (j - 1u < m) {
            const uint32_t ch = inc;
            const int32_t inL = (int32_t)(int16_t)(in & 0xFFFFu), inR = (int32_t)(int16_t)(in >> 16);
            int32_t upL = inL, upR = inR, diagL = dgL, diagR = dgR;
            const bool ch_n = is_n(ch);
            uint32_t packed[(R + 3) / 4];
#pragma unroll
            for (int q = 0; q < (R + 3) / 4; ++q) packed[q] = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int32_t leftL = cL[r], leftG = cG[r], leftR = cR[r];
              const int32_t sub = (rc[r] == ch || ch_n || is_n(rc[r])) ? 1 : -1;
              // L plane: diag, up (1), left (2); strict '<' replaces
              int32_t v = diagL + sub; uint32_t dl = 0;
              if (v < upL - 1) { v = upL - 1; dl = 1; }
              if (v < leftL - 1) { v = leftL - 1; dl = 2; }
              // G plane: stay (2) or enter from L (-2)
              int32_t g = leftG; uint32_t dg = 0;
              if (g < leftL) { g = leftL; dg = 1; }
              // R plane: diag, left (2; free in the last EST row), from G (-2), up (1)
              int32_t rv = diagR + sub; uint32_t dr = 0;
              const int32_t lc = (row0 + r + 1 != n) ? leftR - 1 : leftR;
              if (rv < lc) { rv = lc; dr = 2; }
              if (rv < leftG) { rv = leftG; dr = 3; }
              if (rv < upR - 1) { rv = upR - 1; dr = 1; }
              packed[r / 4] |= (dl | (dg << 2) | (dr << 3)) << (8 * (r % 4));
              diagL = leftL; diagR = leftR;
              cL[r] = v; cG[r] = g; cR[r] = rv;
              upL = v; upR = rv;
            }
            dgL = inL; dgR = inR;
            out = ((uint32_t)upL & 0xFFFFu) | ((uint32_t)upR << 16);
            outc = ch;
            uint8_t* p = dirs + ((size_t)s * 64 + lane) * R;
            if constexpr (R == 1)      *p = (uint8_t)packed[0];
            else if constexpr (R == 2) *reinterpret_cast<uint16_t*>(p) = (uint16_t)packed[0];
            else if constexpr (R == 4) *reinterpret_cast<uint32_t*>(p) = packed[0];
            else if constexpr (R == 8) *reinterpret_cast<uint2*>(p) = make_uint2(packed[0], packed[1]);
            else {
#pragma unroll
              for (int q = 0; q < R / 16; ++q)
                reinterpret_cast<uint4*>(p)[q] =
                    make_uint4(packed[4 * q], packed[4 * q + 1], packed[4 * q + 2], packed[4 * q + 3]);
            }
          }
        }
      }
    }
  }
  // start plane (src/refine-intron.c:808-819); with n==0 or m==0 every plane is 0 -> R
  if (n == 0 || m == 0) {
    if (lane == 0) { res->status = 0; res->pad = 2; }
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r + 1 == n) {
      const int32_t fl = cL[r], fg = cG[r], fr = cR[r];
      int plane;
      if (fr >= fg) plane = fr >= fl ? 2 : 0; else plane = fg >= fl ? 1 : 0;
      res->status = 0;
      res->pad = plane;
    }
  }
}

// all row classes of a batch's gap alignments in one launch (see lev_any_kernel); BIG = 8 rows per
// lane and more (gap alignments of more than 256 EST characters: rare)
template <bool BIG>
__global__ __launch_bounds__(256)
void gap_any_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results,
                    uint8_t* __restrict__ ws, uint8_t* __restrict__ strs) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[4][TB_WIN_BYTES];
  __shared__ uint8_t s_path[4][TB_PATH];
  const uint32_t lane = threadIdx.x & 63u;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= njobs) return;
  const DevJob job = jobs[w];
  DevResult* res = &results[job.out_idx];
  if constexpr (BIG) {
    switch (job.r_class) {
      case 8:  gap_wave_body<8>(job, res, ws, lane); break;
      case 16: gap_wave_body<16>(job, res, ws, lane); break;
      default: gap_wave_body<32>(job, res, ws, lane); break;
    }
  } else {
    switch (job.r_class) {
      case 1:  gap_wave_body<1>(job, res, ws, lane); break;
      case 2:  gap_wave_body<2>(job, res, ws, lane); break;
      default: gap_wave_body<4>(job, res, ws, lane); break;
    }
  }
  own_stores_visible();                  // the planes and the start plane (res->pad)
  gap_traceback_wave(job, res, ws, strs, lane, s_win[threadIdx.x >> 6], s_path[threadIdx.x >> 6]);
}

// TracebackGapAlignment (src/refine-intron.c:828-890), one wave per job: same scheme as
// align_traceback_wave_kernel (direction window in LDS, scalar walk, parallel write-out); the walk
// additionally carries the plane (R exon -> G intron -> L exon) and notes where it jumps.
__device__ __forceinline__ void gap_traceback_wave(const DevJob& job, DevResult* res, const uint8_t* __restrict__ ws,
                                                   uint8_t* __restrict__ strs, const uint32_t lane,
                                                   uint8_t* win, uint8_t* path) {
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.la), m = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.lb), cap = n + m + 1,
                 R = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.r_class);       // uniform walk state: see align_traceback_wave
  const uint32_t lgR = 31u - (uint32_t)__builtin_clz(R);
  const uint32_t WS = TB_WIN_BYTES / (64u * R);            // 1 B per cell: an entry is R bytes
  uint8_t* ea = strs + job.str_off;
  uint8_t* ga = ea + cap;
  const uint8_t* dirs = ws + job.ws_off;
  int plane = __builtin_amdgcn_readfirstlane(res->pad);
  int32_t factor_cut = 0, intron_start = 0, intron_end = 0;
  int32_t rev_end = -1, rev_start = -1;
  uint32_t i = n, j = m, k = 0, np = 0;
  uint32_t i0 = n, j0 = m, pos = cap - 1;
  if (lane == 0) { ea[pos] = 0; ga[pos] = 0; }
  uint32_t s_lo = 1u, s_hi = 0u;
  while (i > 0 && j > 0) {
    const uint32_t l = (i - 1) >> lgR, r = (i - 1) & (R - 1), s = (j - 1) + l;
    if (s < s_lo || s > s_hi) {
      s_hi = s; s_lo = s + 1 >= WS ? s + 1 - WS : 0;
      const uint32_t bytes = (s_hi - s_lo + 1) * 64u * R;
      const uint8_t* src = dirs + (size_t)s_lo * 64u * R;
      for (uint32_t off = lane * 16u; off < bytes; off += 64u * 16u)
        *reinterpret_cast<uint4*>(win + off) = *reinterpret_cast<const uint4*>(src + off);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t b = win[((s - s_lo) * 64u + l) * R + r];
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    // decode to the reference's direction values: 0, 1, 2, or -2 (here 3); selects, not branches
    const uint32_t dR = (b >> 3) & 3u, dG = 2u + ((b >> 2) & 1u), dL = b & 3u;
    const uint32_t d = plane == 2 ? dR : (plane == 1 ? dG : dL);
    path[np] = (uint8_t)d;                                 // every lane, same address, same value
    const uint32_t kk = k + np;                            // steps taken before this one
    ++np;
    if (d == 3) {                                          // twice per job: the walk changes plane
      if (plane == 2) { intron_end = (int32_t)j - 1; factor_cut = (int32_t)i; rev_end = (int32_t)kk; }
      else            { intron_start = (int32_t)j - 1; rev_start = (int32_t)kk; }
      --plane;
    }
    i -= d < 2u ? 1u : 0u;
    j -= d != 1u ? 1u : 0u;
    if (np == TB_PATH) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
      pos -= np; k += np; np = 0; i0 = i; j0 = j;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  tb_flush(path, np, job.a, job.b, i0, j0, pos, ea, ga, lane);
  pos -= np; k += np;
  for (uint32_t q = lane; q < i; q += 64) { ea[pos - 1 - q] = job.a[i - 1 - q]; ga[pos - 1 - q] = '-'; }
  pos -= i; k += i;
  for (uint32_t q = lane; q < j; q += 64) { ea[pos - 1 - q] = '-'; ga[pos - 1 - q] = job.b[j - 1 - q]; }
  pos -= j; k += j;
  if (lane == 0) {
    res->v[0] = (int32_t)k;
    res->v[1] = factor_cut; res->v[2] = intron_start; res->v[3] = intron_end;
    res->v[4] = rev_start >= 0 ? (int32_t)k - 1 - rev_start : 0;
    res->v[5] = rev_end >= 0 ? (int32_t)k - 1 - rev_end : 0;
    res->pad = 0;
    res->str[0] = job.str_off + pos;
    res->str[1] = job.str_off + cap + pos;
  }
}

// ---------------------------------------------------------------------------------------------
// find_longest_common_factor_dp (src/factorization-refinement.c:255-316) from the suffix array.
// The small-exon search asks for the longest common factor of the WHOLE genomic prefix T[0..G) and at most
// 46 characters of the EST (:532): 46 x G cells per call in the reference (and in lcf_kernel), two calls per
// EST -- 9 M cells each on a 200 kb gene, 46 M on a 1 Mb one.  With the index resident the question is
// 64 pattern searches: lane i2 looks for the longest prefix of s2[i2..] that occurs in T and ENDS before G,
//   l <= 8     first_occ_l[code] <= G - l          (a table of first occurrences per l-mer: one probe per l,
//                                                   all eight independent of each other)
//   l  > 8     the interval of the 8-mer in the suffix array, narrowed one character at a time by bisection
//              (a handful of suffixes on a gene-sized sequence); "some occurrence ends before G" is
//              min(SA[lo..hi)) <= G - l, a range minimum read off a sparse table; as soon as one suffix is
//              left the rest is a direct comparison of the two strings.
// "Valid at l + 1" implies "valid at l", so a lane stops at its first failure.  The reference keeps the FIRST
// maximum in (i1, i2) scan order = the longest factor with the smallest start in T, then in s2: every lane
// carries the smallest start of its best length and the wave agrees on (max length, min start, min lane).
// Only exact matching is expressed this way: the library sends a job here when T[0..G) and s2 consist of
// upper-case A, C, G, T alone (no N wildcard can fire: Ns_ALWAYS_MATCH_FOR_LCS, :74), and to lcf_kernel otherwise.
// ---------------------------------------------------------------------------------------------
// find_longest_common_factor_dp of two SHORT strings (the exon ends the small-exon search compares,
// src/factorization-refinement.c:690-718: at most 23 x 23 cells; nine jobs in ten of a C3 batch): one wave,
// lane = diagonal, in rounds of 64 diagonals; N wildcard and first-maximum rule as in lcf_kernel (same key).
// These used to get a workgroup, an atomic and a host-side decode each, in a launch of their own.
__device__ __forceinline__ unsigned long long lcf_key(uint32_t len, uint32_t occ1, uint32_t occ2);
__device__ __forceinline__ void lcf_small_wave_body(const DevJob& job, DevResult* res, const uint32_t lane) {
  const uint32_t l1 = job.la, l2 = job.lb;
  const uint8_t* __restrict__ s1 = job.a; const uint8_t* __restrict__ s2 = job.b;
  unsigned long long best = 0;
  if (l1 != 0 && l2 != 0) {
    const uint32_t ndiag = l1 + l2 - 1;
    for (uint32_t dg0 = 0; dg0 < ndiag; dg0 += 64) {
      const uint32_t dg = dg0 + lane;
      if (dg < ndiag) {
        // cells of this diagonal: i2 from max(0, l2-1-dg), i1 = i2 + dg - (l2-1)
        const uint32_t i2_lo = dg < l2 - 1 ? l2 - 1 - dg : 0;
        const int32_t shift = (int32_t)dg - (int32_t)(l2 - 1);
        uint32_t i2_hi = l2;
        if ((int32_t)l1 - shift < (int32_t)i2_hi) i2_hi = (uint32_t)((int32_t)l1 - shift);
        uint32_t run = 0, brun = 0, bend = 0;
        for (uint32_t i2 = i2_lo; i2 < i2_hi; ++i2) {
          const uint32_t c1 = s1[(int32_t)i2 + shift], c2 = s2[i2];
          run = (c1 == c2 || is_n(c1) || is_n(c2)) ? run + 1 : 0;
          if (run > brun) { brun = run; bend = i2; }
        }
        if (brun > 0) {
          const uint32_t occ2 = bend + 1 - brun;
          const unsigned long long key = lcf_key(brun, (uint32_t)((int32_t)occ2 + shift), occ2);
          best = key > best ? key : best;
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned long long x = __shfl_xor(best, o); best = x > best ? x : best; }
  if (lane == 0) {
    res->status = 0;
    res->v[0] = (int32_t)(best >> 44);
    res->v[1] = best ? (int32_t)(0x0FFFFFFFu - (uint32_t)((best >> 16) & 0x0FFFFFFFu)) : 0;
    res->v[2] = best ? (int32_t)(0xFFFFu - (uint32_t)(best & 0xFFFFu)) : 0;
  }
}

__device__ __forceinline__ uint32_t lcfsa_rmq(const LcfIndexView& ix, uint32_t lo, uint32_t hi) {   // min(sa[lo..hi)), hi > lo
  const uint32_t len = hi - lo;
  const uint32_t j = 31u - (uint32_t)__builtin_clz(len);
  const uint32_t* lv = j == 0 ? ix.sa : ix.rmq + (size_t)(j - 1) * ix.n;
  return min(lv[lo], lv[hi - (1u << j)]);
}

// the search for one string s2 of upper-case A, C, G, T: the wave's key (longest, then smallest start in T, then
// smallest start in s2), the same in every lane; 0 = no common factor
__device__ __forceinline__ unsigned long long lcfsa_wave_key(const uint32_t G, const uint8_t* s2, const uint32_t l2,
                                                             const LcfIndexView& ix, const uint32_t lane) {
  const uint8_t* __restrict__ T = ix.T;
  uint32_t best = 0, bt = 0;
  if (lane < l2) {
    const uint32_t i2 = lane, r = l2 - i2, lim = min(r, 8u);
    uint32_t fo[8], code = 0, off = 0, width = 4;
#pragma unroll
    for (uint32_t l = 1; l <= 8; ++l) {                      // eight independent probes
      fo[l - 1] = 0xFFFFFFFFu;
      if (l <= lim) {
        const int b = base_code(s2[i2 + l - 1]);
        code = code * 4u + (uint32_t)(b < 0 ? 0 : b);
        fo[l - 1] = ix.focc[off + code];
      }
      off += width; width *= 4u;
    }
    bool ok = true;
#pragma unroll
    for (uint32_t l = 1; l <= 8; ++l) {
      if (ok && l <= lim && l <= G && fo[l - 1] <= G - l) { best = l; bt = fo[l - 1]; }
      else ok = false;
    }
    if (best == 8 && r > 8) {
      uint32_t lo = ix.klo[code], hi = ix.khi[code], l = 8;
      while (l < r && hi > lo) {
        if (hi - lo == 1) {
          // one suffix left: lengths l+1 .. min(lcp, G - t) are valid with this start
          const uint32_t t = ix.sa[lo], cap = min(r, G - t);
          uint32_t m = l;
          while (m < cap) {
            // eight characters of each side per round trip
            uint32_t eq = 0;
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) {
              const bool in = m + q < cap;
              const uint32_t a = in ? T[t + m + q] : 0u, b = in ? s2[i2 + m + q] : 1u;
              eq |= (a == b ? 1u : 0u) << q;
            }
            const uint32_t run = (uint32_t)__builtin_ctz(~eq);      // matching characters from m on (at most 8)
            m += run;
            if (run < 8) break;
          }
          if (m > best) { best = m; bt = t; }
          break;
        }
        const uint32_t c = s2[i2 + l];
        // suffixes of [lo, hi) share l characters and are ordered by the next one (the end of T first)
        uint32_t a = lo, z = hi;
        while (a < z) { const uint32_t mid = (a + z) >> 1, p = ix.sa[mid] + l; const uint32_t ch = p < ix.n ? T[p] : 0u; if (ch < c) a = mid + 1; else z = mid; }
        const uint32_t nlo = a;
        z = hi;
        while (a < z) { const uint32_t mid = (a + z) >> 1, p = ix.sa[mid] + l; const uint32_t ch = p < ix.n ? T[p] : 0u; if (ch <= c) a = mid + 1; else z = mid; }
        lo = nlo; hi = a;
        if (lo == hi) break;
        ++l;
        const uint32_t mt = lcfsa_rmq(ix, lo, hi);
        if (l <= G && mt <= G - l) { best = l; bt = mt; } else break;
      }
    }
  }
  // the wave's answer: longest, then smallest start in T, then smallest start in s2 (= lane)
  unsigned long long key = best ? ((unsigned long long)best << 40) | ((unsigned long long)(0x0FFFFFFFu - bt) << 8) | (63u - lane) : 0ull;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned long long x = __shfl_xor(key, o); key = x > key ? x : key; }
  return key;
}

// job.p0 = 0: s2 is upper-case ACGT.  job.p0 = q + 1: s2[q] is an N, the only character of s2 that is not ACGT.
// Ns_ALWAYS_MATCH_FOR_LCS (src/factorization-refinement.c:74) makes that N equal to whatever the genomic sequence
// holds there; the genomic prefix is ACGT only (the host checked), so the factors that run over the N are the exact
// factors of the four strings s2 with A, C, G, T in its place, and the reference's first maximum -- longest, then
// smallest start in T, then in s2 -- is the largest of the four keys.  `scratch`: 64 bytes of the wave's LDS.
__device__ __forceinline__ void lcfsa_wave_body(const DevJob& job, DevResult* res, const LcfIndexView& ix, const uint32_t lane,
                                                uint8_t* scratch) {
  const uint32_t G = job.la, l2 = job.lb;
  unsigned long long key;
  if (job.p0 == 0u) key = lcfsa_wave_key(G, job.b, l2, ix, lane);
  else {
    key = 0ull;
    const uint32_t q = job.p0 - 1u;
    if (lane < l2) scratch[lane] = job.b[lane];
    for (uint32_t c = 0; c < 4u; ++c) {
      if (lane == 0) scratch[q] = (uint8_t)"ACGT"[c];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned long long k = lcfsa_wave_key(G, scratch, l2, ix, lane);
      key = k > key ? k : key;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0) {
    res->status = 0;
    res->v[0] = (int32_t)(key >> 40);
    res->v[1] = key ? (int32_t)(0x0FFFFFFFu - (uint32_t)((key >> 8) & 0x0FFFFFFFu)) : 0;
    res->v[2] = key ? (int32_t)(63u - (uint32_t)(key & 63u)) : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// End-exon alignments (ALIGN jobs with p0 = 1: first exon, p0 = 2: last exon; p1 / p2 = the complexity threshold's
// double bits).  handle_endpoints (src/est-factorizations.c:2127-2301) trims the exon by what the alignment shows at
// its outer end, and the next thing the host asks is the exon check of the TRIMMED exon (KBAND with tail = 1).  The
// wave that just wrote the alignment does both ahead of the host: it walks the outer 64 columns the way the host
// will, takes the trimmed exon as sub-operands and runs the exon check on them.  The answer comes back in the spare
// fields of the ALIGN result TOGETHER with the sub-operands it was computed for (v[2], v[3]: head -- characters of a
// and of b trimmed away; tail -- characters of a and of b kept) and the bound it used; the host files it under the
// question those define and finds it only if its own trimming -- done on the strings, as before -- asks exactly that
// question: a slip in this walk costs a second request, never a different result.  v[4] = ok | dust flags << 1 |
// valid << 3 | bound << 8.  Not attempted (valid = 0): the walk leaves the 64 columns, the exon is dropped, or the
// banded distance would need more than one row per lane.
// ---------------------------------------------------------------------------------------------
// (forceinline: as a call it gave dp_batch_kernel a stack frame -- 96 B of scratch per lane, 114 VGPRs -- and the kernel
// then wrote three times the bytes for the SAME jobs with no such job among them: PMC WRITE_SIZE of dp_batch_kernel over
// 20 000 C3 ESTs 778 -> 2 207 MiB, `tools/pmc_write_ab.sh` over the builds before and after; inlined: 106 VGPRs, no scratch)
__device__ __forceinline__ void endpoint_epilogue(const DevJob& job, DevResult* res, const uint8_t* __restrict__ strs,
                                               uint8_t* __restrict__ ws, const uint32_t lane, DevResult* tmp) {
  const uint32_t n = job.la, m = job.lb;
  const uint32_t dim = (uint32_t)__builtin_amdgcn_readfirstlane(res->v[1]);
  const uint8_t* ea = strs + res->str[0];
  const uint8_t* ga = strs + res->str[1];
  if (dim == 0u) return;
  uint32_t sub_a_off = 0, sub_b_off = 0, sub_la = 0, sub_lb = 0;
  bool valid = false;
  if (job.p0 == 1u) {
    // head (:2163-2196): columns from the left until more than five matches in a row
    const uint32_t c = lane < dim ? lane : dim;
    const uint32_t xe = lane < dim ? ea[c] : 0u, xg = lane < dim ? ga[c] : 0u;
    const unsigned long long eq = __ballot(lane < dim && xe == xg), eg = __ballot(xe == '-'), gg = __ballot(xg == '-');
    uint32_t j = 0, matches = 0, cf = 0, ce = 0;
    bool stop = false;
    const uint32_t lim = dim < 64u ? dim : 64u;
    while (j < lim && !stop) {
      if (matches > 5u) stop = true;
      else {
        if (eq >> j & 1ull) { ++cf; ++ce; ++matches; }
        else { if (!(eg >> j & 1ull)) ++cf; if (!(gg >> j & 1ull)) ++ce; matches = 0; }
        ++j;
      }
    }
    if (stop && cf - matches <= n && ce - matches <= m) {       // (not stopped: the exon is dropped, or the walk goes on beyond the window)
      sub_a_off = cf - matches; sub_b_off = ce - matches; sub_la = n - sub_a_off; sub_lb = m - sub_b_off;
      valid = true;
    }
  } else {
    // tail (:2231-2296): columns from the right until more than ten matches in a row, then the gap columns next to
    // that run are closed by pulling the next character over, as far as the characters agree
    const uint32_t wb = dim > 64u ? dim - 64u : 0u;             // lane t holds column wb + t
    const uint32_t col = wb + lane;
    uint32_t xe = col < dim ? ea[col] : 0u, xg = col < dim ? ga[col] : 0u;
    const unsigned long long eq = __ballot(col < dim && xe == xg), eg = __ballot(xe == '-'), gg = __ballot(xg == '-');
    int j = (int)dim - 1, cf = (int)n - 1, ce = (int)m - 1;
    uint32_t matches = 0;
    bool stop = false, inside = true;
    while (j >= 0 && !stop) {
      if (matches > 10u) stop = true;
      else {
        if (j < (int)wb) { inside = false; break; }
        const uint32_t t = (uint32_t)j - wb;
        if (eq >> t & 1ull) { --cf; --ce; ++matches; }
        else { if (!(eg >> t & 1ull)) --cf; if (!(gg >> t & 1ull)) --ce; matches = 0; }
        --j;
      }
    }
    if (inside) {
      int est_cl = cf + (int)matches, gen_cl = ce + (int)matches;
      uint32_t cursor = (uint32_t)(j + (int)matches + 1);
      bool halt = false;
      // character of a row at column q (a NUL behind the row, as in the host's zero-padded copy)
      auto at = [&](uint32_t v, uint32_t q) -> uint32_t { return q < dim ? (uint32_t)__shfl((int)v, (int)(q - wb)) : 0u; };
      while (!halt && cursor < dim - 1u) {
        const uint32_t ec = at(xe, cursor), gc = at(xg, cursor);
        if (!(ec == '-' || gc == '-')) break;
        uint32_t tr = cursor + 1u;
        if (ec == '-') {
          while (at(xe, tr) == '-') ++tr;
          const uint32_t moved = at(xe, tr);
          if (tr < dim && moved == gc) {
            if (col == cursor) xe = moved;
            if (col == tr) xe = '-';
            ++est_cl; ++gen_cl;
          } else halt = true;
        } else {
          while (at(xg, tr) == '-') ++tr;
          const uint32_t moved = at(xg, tr);
          if (tr < dim && moved == ec) {
            if (col == cursor) xg = moved;
            if (col == tr) xg = '-';
            ++est_cl; ++gen_cl;
          } else halt = true;
        }
        ++cursor;
      }
      if (gen_cl >= 0 && est_cl >= 0 && est_cl < (int)n && gen_cl < (int)m) {
        sub_la = (uint32_t)est_cl + 1u; sub_lb = (uint32_t)gen_cl + 1u;
        valid = true;
      }
    }
  }
  if (!valid || sub_lb == 0u) return;                          // (an exon that is empty on the genomic sequence gets no check)
  // the exon check of the trimmed exon: a = the exon on the genomic sequence, b = on the EST (include/pintron_gpu.h)
  DevJob sub = job;
  sub.a = job.b + sub_b_off; sub.la = sub_lb;
  sub.b = job.a + sub_a_off; sub.lb = sub_la;
  {                                                             // max_edit_for_exon (src/est-factorizations.c:1828-1840), in FP64 like the host
    const double len = (double)sub_lb;
    const double rate = sub_lb > 100u ? 0.030 : (sub_lb > 50u ? 0.035 : 0.040);
    const double c = ceil(len * rate);
    sub.p0 = (uint32_t)(c > 1.0 ? c : 1.0);
  }
  sub.tail = 1u;
  const uint32_t big = sub.la > sub.lb ? sub.la : sub.lb, sml = sub.la > sub.lb ? sub.lb : sub.la;
  if (!((2u * sub.p0 + 1u < big && 2u * sub.p0 + 1u <= 64u) || sml <= 64u)) return;      // more than one row per lane
  if (lane == 0) { tmp->status = 1; tmp->v[0] = 0; tmp->v[1] = 0; tmp->v[2] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  lev_wave_body<1, MODE_KBAND>(sub, tmp, ws, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0 && tmp->status == 0) {
    res->v[2] = (int32_t)(job.p0 == 1u ? sub_a_off : sub_la);
    res->v[3] = (int32_t)(job.p0 == 1u ? sub_b_off : sub_lb);
    res->v[4] = (int32_t)((tmp->v[0] ? 1u : 0u) | (((uint32_t)tmp->v[2] & 3u) << 1) | 8u | (sub.p0 << 8));
  }
}

// ---------------------------------------------------------------------------------------------
// ONE launch for every wave-per-job family of a batch.  A merged batch used to cost eleven launches
// dealt onto four hardware queues, and the kernels of a queue run one after the other: the batch
// took the SUM of its families' long poles per queue.  Here every wave of the grid looks its job up
// in a short segment table (family, first job, count; long-running families first) and runs that
// family's body: alignments with their tracebacks, gap alignments with theirs, banded and plain edit
// distances and the single-wave BORDERS / AFFIX jobs overlap inside one dispatch, and the batch costs
// its longest job.  The rare large row classes (BIG instances, a few hundred registers per lane) and
// the one-job-per-workgroup kernels keep launches of their own.
// ---------------------------------------------------------------------------------------------
struct WaveSegs { int n; int start[MAX_WAVE_SEGS]; int count[MAX_WAVE_SEGS]; int family[MAX_WAVE_SEGS]; };

constexpr size_t WAVE_JOBS_LDS = 4 * (size_t)TB_WIN_BYTES + 4 * (size_t)TB_PATH + 4 * 4 * 65 * sizeof(uint32_t);
static_assert(WAVE_JOBS_LDS + 256 <= 40 * 1024, "four workgroups of dp_batch_kernel per CU (160 KB of LDS; 256 B are static)");

// wave `wave` (0..3) of workgroup `block` of the wave-per-job part; smem: WAVE_JOBS_LDS bytes, 16-aligned
__device__ __forceinline__ void wave_jobs_body(const int block, const int wave, const uint32_t lane,
                                               const DevJob* __restrict__ jobs, const WaveSegs& segs,
                                               DevResult* __restrict__ results, uint8_t* __restrict__ ws,
                                               uint8_t* __restrict__ strs, uint8_t* smem, const LcfIndexView& ix) {
  uint8_t* s_win = smem + (size_t)wave * TB_WIN_BYTES;
  uint8_t* s_path = smem + 4 * (size_t)TB_WIN_BYTES + (size_t)wave * TB_PATH;
  uint32_t* s_borders = reinterpret_cast<uint32_t*>(smem + 4 * (size_t)TB_WIN_BYTES + 4 * (size_t)TB_PATH) + wave * (4 * 65);
  int w = block * 4 + wave, fam = -1, idx = 0;
  for (int sgi = 0; sgi < segs.n; ++sgi) {
    if (w < segs.count[sgi]) { fam = segs.family[sgi]; idx = segs.start[sgi] + w; break; }
    w -= segs.count[sgi];
  }
  if (fam < 0) return;
  const DevJob job = jobs[idx];
  DevResult* res = &results[job.out_idx];
  switch (fam) {
    case KF_ALIGN:
      lev_any_dispatch<MODE_ALIGN, false>(job, res, ws, lane);
      own_stores_visible();
      align_traceback_wave(job, res, ws, strs, lane, s_win, s_path);
      if (job.p0 != 0u) { own_stores_visible(); endpoint_epilogue(job, res, strs, ws, lane, reinterpret_cast<DevResult*>(s_borders)); }
      break;
    case KF_GAP:
      switch (job.r_class) {
        case 1:  gap_wave_body<1>(job, res, ws, lane); break;
        case 2:  gap_wave_body<2>(job, res, ws, lane); break;
        default: gap_wave_body<4>(job, res, ws, lane); break;
      }
      own_stores_visible();
      gap_traceback_wave(job, res, ws, strs, lane, s_win, s_path);
      break;
    case KF_KBAND:   lev_any_dispatch<MODE_KBAND, false>(job, res, ws, lane); break;
    case KF_ED:      lev_any_dispatch<MODE_ED, false>(job, res, ws, lane); break;
    case KF_BORDERS: lev_wave_body<1, MODE_BORDERS>(job, res, ws, lane, s_borders); break;
    case KF_AFFIX:   lev_wave_body<1, MODE_AFFIX>(job, res, ws, lane); break;
    case KF_LCFSA:   lcfsa_wave_body(job, res, ix, lane, reinterpret_cast<uint8_t*>(s_borders)); break;
    case KF_LCFW:    lcf_small_wave_body(job, res, lane); break;
    case KF_ALIGNB: {                    // see align_band_sweep; the host chose the jobs (more than 64 rows, lengths within the band)
      const uint32_t n = job.la, m = job.lb;
      bool same = n == m;
      if (same) for (uint32_t q = lane; q < n; q += 64) same = same && job.a[q] == job.b[q];
      if (__all(same)) {                 // identity alignment, score 0 (compute-alignments.c:48-58)
        if (lane == 0) { res->status = 0; res->v[0] = 0; res->v[1] = (int32_t)n; res->v[5] = 1; }
        own_stores_visible();
        align_traceback_wave(job, res, ws, strs, lane, s_win, s_path);      // its identity branch
        if (job.p0 != 0u) { own_stores_visible(); endpoint_epilogue(job, res, strs, ws, lane, reinterpret_cast<DevResult*>(s_borders)); }
        break;
      }
      uint32_t* bdirs = reinterpret_cast<uint32_t*>(ws + job.ws_off);
      const uint32_t score = align_band_sweep(job.a, n, job.b, m, lane, bdirs);
      if (score > ALIGN_BAND_K) { if (lane == 0) res->status = ALIGN_BAND_RETRY; break; }
      if (lane == 0) { res->status = 0; res->v[0] = (int32_t)score; res->v[5] = 0; }
      own_stores_visible();
      align_band_traceback(job, res, bdirs, strs, lane, s_win, s_path);
      if (job.p0 != 0u) { own_stores_visible(); endpoint_epilogue(job, res, strs, ws, lane, reinterpret_cast<DevResult*>(s_borders)); }
      break;
    }
    default: break;
  }
}

// follow-up of the merged launch: the banded ALIGN jobs whose score exceeded the band, on four waves each;
// a workgroup whose job is settled (nearly all) ends at once
__global__ __launch_bounds__(256)
void align_fallback_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results,
                           uint8_t* __restrict__ ws, uint8_t* __restrict__ strs) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[ALIGN_COOP_LDS];
  DevJob job = jobs[blockIdx.x];
  if (results[job.out_idx].status != ALIGN_BAND_RETRY) return;
  job.tail = 0;                          // the band has been tried
  align_coop_dispatch(job, &results[job.out_idx], ws, strs, smem);
}

__global__ __launch_bounds__(256)
void wave_jobs_kernel(const DevJob* __restrict__ jobs, const WaveSegs segs, DevResult* __restrict__ results,
                      uint8_t* __restrict__ ws, uint8_t* __restrict__ strs, const LcfIndexView ix) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[WAVE_JOBS_LDS];
  const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  wave_jobs_body((int)blockIdx.x, wave, threadIdx.x & 63u, jobs, segs, results, ws, strs, smem, ix);
}

// ---------------------------------------------------------------------------------------------
// ONE launch for everything of a batch that is bound by the latency of its longest job: the
// one-job-per-workgroup sweeps (BORDERS on eight waves, AFFIX on four) and the wave-per-job
// families.  Workgroups of 512 threads; the role of a workgroup follows from its index -- the long
// poles first, so they start first: [BORDERS coop jobs][AFFIX coop jobs][wave-job groups of four].
// The roles with four waves let the upper four end at once (s_barrier only waits for the waves of
// a workgroup that have not ended).  All roles share the dynamic LDS.
// ---------------------------------------------------------------------------------------------
struct BatchDesc { WaveSegs segs; int wave_blocks; int bc_start, bc_count, ac_start, ac_count, lc_start, lc_count; LcfIndexView ix; };

__global__ __launch_bounds__(512)
void dp_batch_kernel(const DevJob* __restrict__ jobs, const BatchDesc d, DevResult* __restrict__ results,
                     uint8_t* __restrict__ ws, uint8_t* __restrict__ strs) {
  extern __shared__ __attribute__((aligned(16))) uint8_t batch_lds[];
  int b = (int)blockIdx.x;
  if (b < d.bc_count) {
    const DevJob job = jobs[d.bc_start + b];
    borders_coop_dispatch(job, &results[job.out_idx]);      // its LDS is the dynamic array
    return;
  }
  b -= d.bc_count;
  if (threadIdx.x >= 256) return;
  if (b < d.ac_count) {
    const DevJob job = jobs[d.ac_start + b];
    uint32_t* hand = reinterpret_cast<uint32_t*>(batch_lds);
    uint32_t (*wbest)[5] = reinterpret_cast<uint32_t (*)[5]>(batch_lds + (COOP_W - 1) * 128 * sizeof(uint32_t));
    affix_coop_dispatch(job, &results[job.out_idx], hand, wbest);
    return;
  }
  b -= d.ac_count;
  if (b < d.lc_count) {
    const DevJob job = jobs[d.lc_start + b];
    align_coop_dispatch(job, &results[job.out_idx], ws, strs, batch_lds);
    return;
  }
  b -= d.lc_count;
  const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  wave_jobs_body(b, wave, threadIdx.x & 63u, jobs, d.segs, results, ws, strs, batch_lds, d.ix);
}
constexpr size_t AFFIX_COOP_LDS = (COOP_W - 1) * 128 * sizeof(uint32_t) + COOP_W * 5 * sizeof(uint32_t);

// ---------------------------------------------------------------------------------------------
// Longest common factor with N wildcard: find_longest_common_factor_dp
// (src/factorization-refinement.c:255-316).  curr[i2+1] = match ? prev[i2]+1 : 0 only couples
// cells of one diagonal, so the l1+l2-1 diagonals are independent: one thread per diagonal,
// 256 diagonals per workgroup, a tile of s1 and the whole of s2 staged in LDS.  The reference
// keeps the FIRST maximum in (i1,i2) scan order == among maximal runs the smallest start in s1,
// then in s2; that order is folded into a 64-bit key reduced with atomicMax.
// ---------------------------------------------------------------------------------------------
constexpr int LCF_BLOCK = 256;
constexpr uint32_t LCF_MAX_L2 = 65535u;

__device__ __forceinline__ unsigned long long lcf_key(uint32_t len, uint32_t occ1, uint32_t occ2) {
  return ((unsigned long long)len << 44) | ((unsigned long long)(0x0FFFFFFFu - occ1) << 16) |
         (unsigned long long)(0xFFFFu - occ2);
}

// grid (workgroups per job, jobs); keys[job] (zeroed by the caller) collects the best key of the job's
// workgroups.  The host turns the keys into results (pgpu_dp_plan_sync): a finish pass on the device
// would cost a launch per batch, and "the last workgroup finishes" needs device-scope fences, i.e. a
// write-back and an invalidation of the XCD's L2 per workgroup -- measured: six times slower, and every
// kernel running beside it with it.
__global__ __launch_bounds__(LCF_BLOCK)
void lcf_kernel(const DevJob* __restrict__ jobs, int njobs, unsigned long long* __restrict__ keys) {
  extern __shared__ uint8_t lcf_lds[];         // [s2: l2 bytes][s1 tile: LCF_BLOCK + l2 bytes]
  __shared__ unsigned long long wbest[LCF_BLOCK / 64];
  const DevJob job = jobs[blockIdx.y];
  const uint32_t l1 = job.la, l2 = job.lb;
  unsigned long long best = 0;
  if (l1 != 0 && l2 != 0) {                    // uniform over the workgroup
    uint8_t* s2 = lcf_lds;
    uint8_t* tile = lcf_lds + ((l2 + 15u) & ~15u);
    for (uint32_t i = threadIdx.x; i < l2; i += LCF_BLOCK) s2[i] = job.b[i];
    const uint32_t ndiag = l1 + l2 - 1;
    const uint32_t nchunks = (ndiag + LCF_BLOCK - 1) / LCF_BLOCK;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
      // diagonal index dg in [0, ndiag): i1 - i2 = dg - (l2-1).  This chunk covers s1 positions
      // [base, base + LCF_BLOCK + l2 - 1) where base = chunk*LCF_BLOCK - (l2-1) (may be negative)
      const int64_t base = (int64_t)chunk * LCF_BLOCK - (int64_t)(l2 - 1);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < LCF_BLOCK + l2 - 1; i += LCF_BLOCK) {
        const int64_t g = base + i;
        tile[i] = (g >= 0 && g < (int64_t)l1) ? job.a[g] : 0;
      }
      __syncthreads();
      const uint32_t dg = chunk * LCF_BLOCK + threadIdx.x;
      if (dg < ndiag) {
        // cells of this diagonal: i2 from max(0, l2-1-dg), i1 = i2 + dg - (l2-1)
        const uint32_t i2_lo = dg < l2 - 1 ? l2 - 1 - dg : 0;
        const int64_t shift = (int64_t)dg - (int64_t)(l2 - 1);       // i1 - i2
        uint32_t i2_hi = l2;                                          // exclusive
        if ((int64_t)l1 - shift < (int64_t)i2_hi) i2_hi = (uint32_t)((int64_t)l1 - shift);
        uint32_t run = 0, brun = 0, bend = 0;
        for (uint32_t i2 = i2_lo; i2 < i2_hi; ++i2) {
          const uint32_t c1 = tile[threadIdx.x + i2];                // s1[i2 + shift] = tile[i2+shift-base]
          const uint32_t c2 = s2[i2];
          run = (c1 == c2 || is_n(c1) || is_n(c2)) ? run + 1 : 0;
          if (run > brun) { brun = run; bend = i2; }
        }
        if (brun > 0) {
          const uint32_t occ2 = bend + 1 - brun;
          const uint32_t occ1 = (uint32_t)((int64_t)occ2 + shift);
          const unsigned long long key = lcf_key(brun, occ1, occ2);
          best = key > best ? key : best;
        }
      }
    }
    // workgroup reduction, one atomic per workgroup
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(best, off);
      best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (l1 != 0 && l2 != 0) {
      for (int wv = 1; wv < LCF_BLOCK / 64; ++wv) best = wbest[wv] > best ? wbest[wv] : best;
      if (best) atomicMax(&keys[blockIdx.y], best);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Jobs beyond the row limits of the register-resident kernels (GAP windows of more than 2048 EST
// characters, BORDERS patterns of more than 4096): the reference computes them, slowly, so they
// must not be refused.  One workgroup per job walks the anti-diagonals of the matrix with three
// rolling diagonals per plane in the job's HBM workspace; cell (i, j) only needs diagonals d-1 and
// d-2, thread q owns the rows i = q+1, q+1+256, ...  Nothing here is tuned: such jobs are rare
// (long reads with a long unaligned stretch) and a slow answer beats an abort.
// ---------------------------------------------------------------------------------------------
constexpr int SLOW_BLOCK = 256;

// ComputeGapAlignMatrix + TracebackGapAlignment (src/refine-intron.c:623-890), semantics as in
// gap_wave_body / gap_traceback_wave.  Workspace: [9 x (n+1) int32 rolling diagonals][(n+1)*(m+1) bytes:
// bits 0-1 L direction, bit 2 G "came from L", bits 3-4 R direction (3 = came from G)]
__global__ __launch_bounds__(SLOW_BLOCK)
void gap_slow_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results,
                     uint8_t* __restrict__ ws, uint8_t* __restrict__ strs) {
  const DevJob job = jobs[blockIdx.x];
  DevResult* res = &results[job.out_idx];
  const uint32_t n = job.la, m = job.lb, cap = n + m + 1;
  int32_t* diag = (int32_t*)(ws + job.ws_off);
  uint8_t* dirs = ws + job.ws_off + (size_t)9 * (n + 1) * sizeof(int32_t);
  const size_t w = (size_t)m + 1;
  auto plane = [&](int which, uint32_t d) -> int32_t* { return diag + ((size_t)(which * 3 + (int)(d % 3u))) * (n + 1); };
  for (uint32_t d = 2; d <= n + m; ++d) {
    const int32_t *L1 = plane(0, d - 1), *L2 = plane(0, d - 2), *G1 = plane(1, d - 1), *R1 = plane(2, d - 1), *R2 = plane(2, d - 2);
    int32_t *Lc = plane(0, d), *Gc = plane(1, d), *Rc = plane(2, d);
    for (uint32_t i = threadIdx.x + 1; i <= n; i += SLOW_BLOCK) {
      if (d <= i) break;
      const uint32_t j = d - i;
      if (j > m) continue;
      // neighbours on the zero border (row 0 / column 0 of every plane) read as 0
      const bool up_ok = i > 1, left_ok = j > 1;
      const uint32_t ce = job.a[i - 1], cg = job.b[j - 1];
      const int32_t sub = (ce == cg || is_n(ce) || is_n(cg)) ? 1 : -1;
      const int32_t l_diag = (up_ok && left_ok) ? L2[i - 1] : 0, l_up = up_ok ? L1[i - 1] : 0, l_left = left_ok ? L1[i] : 0;
      const int32_t g_left = left_ok ? G1[i] : 0;
      const int32_t r_diag = (up_ok && left_ok) ? R2[i - 1] : 0, r_up = up_ok ? R1[i - 1] : 0, r_left = left_ok ? R1[i] : 0;
      int32_t v = l_diag + sub; uint32_t dl = 0;
      if (v < l_up - 1) { v = l_up - 1; dl = 1; }
      if (v < l_left - 1) { v = l_left - 1; dl = 2; }
      Lc[i] = v;
      v = g_left; uint32_t dg = 0;
      if (v < l_left) { v = l_left; dg = 1; }
      Gc[i] = v;
      v = r_diag + sub; uint32_t dr = 0;
      const int32_t left = (i != n) ? r_left - 1 : r_left;            // free trailing gap (:756-759)
      if (v < left) { v = left; dr = 2; }
      if (v < g_left) { v = g_left; dr = 3; }
      if (v < r_up - 1) { v = r_up - 1; dr = 1; }
      Rc[i] = v;
      dirs[(size_t)i * w + j] = (uint8_t)(dl | (dg << 2) | (dr << 3));
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  uint8_t* ea = strs + job.str_off;
  uint8_t* ga = ea + cap;
  int32_t pl = 0;
  if (n > 0 && m > 0) {
    const uint32_t d = n + m;
    const int32_t l = plane(0, d)[n], g = plane(1, d)[n], r = plane(2, d)[n];
    if (r >= g) pl = (r >= l) ? 2 : 0; else pl = (g >= l) ? 1 : 0;   // :808-819
  } else pl = 2;                                                   // all three are 0
  int32_t factor_cut = 0, intron_start = 0, intron_end = 0, rev_start = -1, rev_end = -1;
  uint32_t i = n, j = m, pos = cap - 1, k = 0;
  ea[pos] = 0; ga[pos] = 0;
  while (i > 0 && j > 0) {
    const uint32_t b = dirs[(size_t)i * w + j];
    const uint32_t dd = pl == 2 ? ((b >> 3) & 3u) : (pl == 1 ? (((b >> 2) & 1u) ? 3u : 2u) : (b & 3u));
    --pos;
    if (dd == 0)      { ea[pos] = job.a[--i]; ga[pos] = job.b[--j]; }
    else if (dd == 1) { ea[pos] = job.a[--i]; ga[pos] = '-'; }
    else {
      if (dd == 3) {
        if (pl == 2) { intron_end = (int32_t)j - 1; factor_cut = (int32_t)i; rev_end = (int32_t)k; }
        else         { intron_start = (int32_t)j - 1; rev_start = (int32_t)k; }
        --pl;
      }
      ea[pos] = '-'; ga[pos] = job.b[--j];
    }
    ++k;
  }
  while (i > 0) { --pos; ea[pos] = job.a[--i]; ga[pos] = '-'; ++k; }
  while (j > 0) { --pos; ea[pos] = '-'; ga[pos] = job.b[--j]; ++k; }
  res->status = 0;
  res->v[0] = (int32_t)k; res->v[1] = factor_cut; res->v[2] = intron_start; res->v[3] = intron_end;
  res->v[4] = rev_start >= 0 ? (int32_t)k - 1 - rev_start : 0;
  res->v[5] = rev_end >= 0 ? (int32_t)k - 1 - rev_end : 0;
  res->pad = 0;
  res->str[0] = job.str_off + pos;
  res->str[1] = job.str_off + cap + pos;
}

// general_refine_borders (src/refine.c:105-192) for patterns of more than 4096 characters.
// Workspace: [3 x (len_p+1) uint32 rolling diagonals][pre, pre_pos, suf, suf_pos: 4 x (len_p+1) uint32]
__global__ __launch_bounds__(SLOW_BLOCK)
void borders_slow_kernel(const DevJob* __restrict__ jobs, int njobs, DevResult* __restrict__ results, uint8_t* __restrict__ ws) {
  const DevJob job = jobs[blockIdx.x];
  DevResult* res = &results[job.out_idx];
  const uint32_t len_p = job.la, len_t = job.lb, max_errs = job.p2;
  const uint32_t t_win = min(len_p + max_errs, len_t);
  uint32_t* diag = (uint32_t*)(ws + job.ws_off);
  uint32_t* mins = diag + (size_t)3 * (len_p + 1);
  for (int sweep = 0; sweep < 2; ++sweep) {
    uint32_t* mv = mins + (size_t)(2 * sweep) * (len_p + 1);
    uint32_t* mp = mv + (len_p + 1);
    const Operand rows{job.a, len_p, sweep == 1}, cols{job.b, len_t, sweep == 1};
    for (uint32_t i = threadIdx.x; i <= len_p; i += SLOW_BLOCK) { mv[i] = i; mp[i] = 0; }     // column 0: M[i][0] = i
    __syncthreads();
    for (uint32_t d = 2; d <= len_p + t_win; ++d) {
      const uint32_t *D1 = diag + (size_t)((d - 1) % 3u) * (len_p + 1), *D2 = diag + (size_t)((d - 2) % 3u) * (len_p + 1);
      uint32_t* Dc = diag + (size_t)(d % 3u) * (len_p + 1);
      for (uint32_t i = threadIdx.x + 1; i <= len_p; i += SLOW_BLOCK) {
        if (d <= i) break;
        const uint32_t j = d - i;
        if (j > t_win) continue;
        const uint32_t dg = i == 1 ? j - 1 : (j == 1 ? i - 1 : D2[i - 1]);
        const uint32_t up = i == 1 ? j : D1[i - 1];
        const uint32_t lf = j == 1 ? i : D1[i];
        uint32_t v = dg + (rows.at(i - 1) == cols.at(j - 1) ? 0u : 1u);
        if (v > up + 1) v = up + 1;
        if (v > lf + 1) v = lf + 1;
        Dc[i] = v;
        if (mv[i] > v) { mv[i] = v; mp[i] = j; }            // strict: the first arg-min of the row (:133-159)
      }
      __syncthreads();
    }
  }
  if (threadIdx.x != 0) return;
  const uint32_t *pre = mins, *pre_pos = mins + (len_p + 1), *suf = mins + (size_t)2 * (len_p + 1), *suf_pos = mins + (size_t)3 * (len_p + 1);
  const uint32_t avail = len_t + min(job.tail, 2u);
  const uint32_t lo = job.p0, hi = job.p1 > job.p0 ? job.p1 : job.p0;
  uint32_t bi = lo, bc = pre[lo] + suf[len_p - lo];
  int bf = burset_adaptor(job.b, avail, pre_pos[lo], len_t - suf_pos[len_p - lo]);
  for (uint32_t i = lo + 1; i <= hi; ++i) {
    const int freq = burset_adaptor(job.b, avail, pre_pos[i], len_t - suf_pos[len_p - i]);
    const uint32_t c = pre[i] + suf[len_p - i];
    if (bc > c || (bc == c && freq > bf)) { bc = c; bf = freq; bi = i; }
  }
  res->status = 0;
  res->v[0] = bc <= max_errs ? 1 : 0;
  res->v[1] = (int32_t)bi; res->v[2] = (int32_t)pre_pos[bi];
  res->v[3] = (int32_t)(len_t - suf_pos[len_p - bi]); res->v[4] = (int32_t)bc;
}

}  // namespace

// R = 0: every row class of the family in one launch (ED, ALIGN, KBAND: the first n_big jobs are
// of the classes above 16 rows per lane and go to the BIG instance; BORDERS / AFFIX above 64
// rows, `max_rows` sizes the dynamic LDS); R = 1: the single-wave BORDERS / AFFIX kernels;
// R = ROW_CLASS_STRIPS: AFFIX beyond 4096 rows.
template <int MODE>
static void launch_lev_any(const DevJob* jobs, int njobs, int n_big, DevResult* res, uint8_t* ws, uint8_t* strs, hipStream_t st) {
  const dim3 b256(256);
  if (n_big > 0)
    hipLaunchKernelGGL((lev_any_kernel<MODE, true>), dim3((n_big + 3) / 4), b256, 0, st, jobs, n_big, res, ws, strs);
  if (njobs > n_big)
    hipLaunchKernelGGL((lev_any_kernel<MODE, false>), dim3((njobs - n_big + 3) / 4), b256, 0, st, jobs + n_big, njobs - n_big, res, ws, strs);
}

void launch_lev(int family, int R, uint32_t max_rows, const DevJob* jobs, int njobs, int n_big, DevResult* res, uint8_t* ws,
                uint8_t* strs, hipStream_t st) {
  if (njobs <= 0) return;
  const dim3 g4((njobs + 3) / 4), b256(256);
  switch (family) {
    case KF_ED:    launch_lev_any<MODE_ED>(jobs, njobs, n_big, res, ws, strs, st); break;
    case KF_ALIGN:
      if (R == 0) hipLaunchKernelGGL(align_coop_any_kernel, dim3(njobs), dim3(256), 0, st, jobs, njobs, res, ws, strs);   // 65 .. 4096 rows
      else launch_lev_any<MODE_ALIGN>(jobs, njobs, n_big, res, ws, strs, st);
      break;
    case KF_KBAND: launch_lev_any<MODE_KBAND>(jobs, njobs, n_big, res, ws, strs, st); break;
    case KF_BORDERS:
      if (R == 1) {
        const size_t lds = 4 * (64 + 1) * sizeof(uint32_t);
        hipLaunchKernelGGL((lev_wave_kernel<1, MODE_BORDERS>), dim3(njobs), dim3(64), lds, st, jobs, njobs, res, ws);
      } else if (R == (int)ROW_CLASS_STRIPS) {
        hipLaunchKernelGGL(borders_slow_kernel, dim3(njobs), dim3(SLOW_BLOCK), 0, st, jobs, njobs, res, ws);
      } else {
        const size_t lds = (2 * (COOP_W - 1) * 128 + 4 * ((size_t)max_rows + 1)) * sizeof(uint32_t);
        hipLaunchKernelGGL(borders_coop_any_kernel, dim3(njobs), dim3(512), lds, st, jobs, njobs, res);
      }
      break;
    case KF_AFFIX:
      if (R == 1) hipLaunchKernelGGL((lev_wave_kernel<1, MODE_AFFIX>), g4, b256, 0, st, jobs, njobs, res, ws);
      else if (R == (int)ROW_CLASS_STRIPS) hipLaunchKernelGGL((lev_wave_kernel<64, MODE_AFFIX, true>), g4, b256, 0, st, jobs, njobs, res, ws);
      else hipLaunchKernelGGL(affix_coop_any_kernel, dim3(njobs), dim3(256), 0, st, jobs, njobs, res);
      break;
    default: break;
  }
}

void launch_align_fallback(const DevJob* jobs, int njobs, DevResult* res, uint8_t* ws, uint8_t* strs, hipStream_t st) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(align_fallback_kernel, dim3(njobs), dim3(256), 0, st, jobs, njobs, res, ws, strs);
}

// segments: (family, first job, count) of the jobs the merged kernel runs, long poles first
void launch_wave_jobs(const DevJob* jobs, int n_segs, const int* family, const int* start, const int* count,
                      DevResult* res, uint8_t* ws, uint8_t* strs, const LcfIndexView& ix, hipStream_t st) {
  WaveSegs sg;
  sg.n = 0;
  int total = 0;
  for (int k = 0; k < n_segs && sg.n < MAX_WAVE_SEGS; ++k) {
    if (count[k] <= 0) continue;
    sg.family[sg.n] = family[k]; sg.start[sg.n] = start[k]; sg.count[sg.n] = count[k]; ++sg.n;
    total += count[k];
  }
  if (total == 0) return;
  hipLaunchKernelGGL(wave_jobs_kernel, dim3((total + 3) / 4), dim3(256), 0, st, jobs, sg, res, ws, strs, ix);
}

void launch_gap_slow(const DevJob* jobs, int njobs, DevResult* res, uint8_t* ws, uint8_t* strs, hipStream_t st) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(gap_slow_kernel, dim3(njobs), dim3(SLOW_BLOCK), 0, st, jobs, njobs, res, ws, strs);
}

void launch_gap(const DevJob* jobs, int njobs, int n_big, DevResult* res, uint8_t* ws, uint8_t* strs, hipStream_t st) {
  if (njobs <= 0) return;
  if (n_big > 0)
    hipLaunchKernelGGL(gap_any_kernel<true>, dim3((n_big + 3) / 4), dim3(256), 0, st, jobs, n_big, res, ws, strs);
  if (njobs > n_big)
    hipLaunchKernelGGL(gap_any_kernel<false>, dim3((njobs - n_big + 3) / 4), dim3(256), 0, st, jobs + n_big, njobs - n_big, res, ws, strs);
}

void launch_lcf(const DevJob* jobs, int njobs, uint32_t max_chunks, uint32_t max_l2,
                unsigned long long* keys, hipStream_t st) {
  if (njobs <= 0) return;
  // grid.y = job (<= 65535 per launch, the caller slices), grid.x strides over diagonal chunks
  // ... as many workgroups per job as it takes to fill the chip: after the suffix-array path took the ordinary
  // jobs, what comes here is a handful of long ones (an N in the EST prefix) -- 64 workgroups each left three
  // quarters of the CUs idle and made these launches the long pole of their batches
  uint32_t want = 4096u / (uint32_t)njobs;
  if (want < 64u) want = 64u;
  const uint32_t gx = max_chunks < want ? (max_chunks ? max_chunks : 1u) : want;
  // LDS: s2 (rounded to 16) + s1 tile (256 + l2 - 1), sized for the largest l2 of the launch
  const size_t lds = ((max_l2 + 15u) & ~15u) + LCF_BLOCK + max_l2;
  hipLaunchKernelGGL(lcf_kernel, dim3(gx, njobs), dim3(LCF_BLOCK), lds, st, jobs, njobs, keys);
}

constexpr size_t DP_BATCH_MAX_LDS = 64 * 1024;

size_t dp_batch_lds_bytes(bool wave_jobs, int bc_count, uint32_t bc_max_rows, int ac_count, int lc_count) {
  size_t lds = 16;
  if (wave_jobs) lds = std::max(lds, WAVE_JOBS_LDS);
  if (ac_count > 0) lds = std::max(lds, AFFIX_COOP_LDS);
  if (lc_count > 0) lds = std::max(lds, ALIGN_COOP_LDS);
  if (bc_count > 0) lds = std::max(lds, (2 * (COOP_W - 1) * 128 + 4 * ((size_t)bc_max_rows + 1)) * sizeof(uint32_t));
  return lds;
}

bool launch_dp_batch(const DevJob* jobs, int n_segs, const int* family, const int* start, const int* count,
                     int bc_start, int bc_count, uint32_t bc_max_rows, int ac_start, int ac_count, int lc_start, int lc_count,
                     DevResult* res, uint8_t* ws, uint8_t* strs, const LcfIndexView& ix, hipStream_t st) {
  BatchDesc d;
  d.ix = ix;
  d.lc_start = lc_start; d.lc_count = lc_count > 0 ? lc_count : 0;
  d.segs.n = 0;
  int total = 0;
  for (int k = 0; k < n_segs && d.segs.n < MAX_WAVE_SEGS; ++k) {
    if (count[k] <= 0) continue;
    d.segs.family[d.segs.n] = family[k]; d.segs.start[d.segs.n] = start[k]; d.segs.count[d.segs.n] = count[k]; ++d.segs.n;
    total += count[k];
  }
  d.wave_blocks = (total + 3) / 4;
  d.bc_start = bc_start; d.bc_count = bc_count > 0 ? bc_count : 0;
  d.ac_start = ac_start; d.ac_count = ac_count > 0 ? ac_count : 0;
  const int blocks = d.bc_count + d.ac_count + d.lc_count + d.wave_blocks;
  if (blocks == 0) return true;
  static const size_t lds_pad = [] { const char* e = getenv("PGPU_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();   // measurement only
  const size_t lds = dp_batch_lds_bytes(total > 0, d.bc_count, bc_max_rows, d.ac_count, d.lc_count) + lds_pad;
  if (lds > DP_BATCH_MAX_LDS) return false;
  hipLaunchKernelGGL(dp_batch_kernel, dim3(blocks), dim3(512), lds, st, jobs, d, res, ws, strs);
  return true;
}
