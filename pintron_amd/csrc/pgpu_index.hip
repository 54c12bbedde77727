// Genomic index resident in HBM -- the sequence (1 B/base), its suffix array and LCP array -- and
// the pairing kernels that replace the reference's augmented suffix tree:
//   lst_stree_new (stree_src/lst_stree.c:816), stree_preprocess (src/aug_suffix_tree.c:247) and
//   build_vertex_set (src/max-emb-graph.c:218-392).
// Semantics (how the tree walk maps onto SA intervals, the suffix-link start rule, the two
// low-complexity filters) are stated in DESIGN.md section 4b and restated on the CPU, for the
// tests only, in oracle/pairing_oracle.c.
//
// Index build: prefix doubling on the device; the device-wide sorts and scans are rocPRIM
// (one-off per gene, not a hot path).  Everything per-EST is hand-written:
//   pair_locate   one wave per pattern, one lane per position: SA interval of the L-mer
//                 (two binary searches, T and SA are L2-resident: 9 B/base), then the longest
//                 match among the occurrences that are not preceded by P[i-1]
//   pair_chain    one thread per pattern: the sequential locus-depth recurrence
//                 D_i = max(A_i, s_i), threshold, next start depth from LCP-interval borders
//   pair_count / pair_fill / pair_cross / pair_emit
//                 per position: occurrences above the threshold, sort by t, filter (a);
//                 filter (b) against the previous position; compaction into (p,t,l) triples
#include <algorithm>
#include <unistd.h>
#include <sys/stat.h>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>   // index construction only (one-off per gene)
#include <rocprim/device/device_scan.hpp>

#include "pgpu_index.h"

// The first KTAB characters of a pattern position select the suffix-array interval of that
// KTAB-mer from a table (2 x 4^KTAB x 4 B = 512 KB, L2-resident) instead of ~2 x 17 bisection
// steps over the whole array; the bisection continues inside that interval (a handful of
// suffixes) from character KTAB on.  Only upper-case ACGT k-mers are tabulated.
constexpr uint32_t KTAB = 8;
constexpr uint32_t KTAB_ENTRIES = 1u << (2 * KTAB);

struct pgpu_index {
  uint8_t* d_gen = nullptr;
  uint32_t* d_sa = nullptr;
  uint32_t* d_lcp = nullptr;      // n+1 entries; lcp[0] = lcp[n] = 0
  uint32_t* d_klo = nullptr;      // [code] -> first suffix-array index of the k-mer (klo == khi: absent)
  uint32_t* d_khi = nullptr;
  // preprocess_text (src/aug_suffix_tree.c:68-120): key of a character = its rank among the distinct
  // characters of the genomic sequence, `sigma` (their number) for characters that do not occur
  uint8_t* d_key = nullptr;          // 256 entries
  uint32_t sigma = 0;
  size_t len = 0;
  // derived tables for the suffix-array longest-common-factor search (LcfIndexView, pgpu_index.h); made
  // after every build and load, not kept in the index file
  uint32_t* d_focc = nullptr;
  uint32_t* d_rmq = nullptr;
  uint32_t rmq_levels = 0, first_bad = 0;
  // the sequence at 2 bits per base (16 bases per word) + 1 flag bit per base for "not an upper-case A, C, G, T"
  // (32 per word): what the pairing kernels compare on (PackedSeq); the bytes stay for the flagged stretches
  uint32_t* d_code = nullptr;
  uint32_t* d_bad = nullptr;
};

LcfIndexView pgpu_index_lcf_view(const pgpu_index* idx) {
  LcfIndexView v{};
  if (idx) { v.T = idx->d_gen; v.sa = idx->d_sa; v.klo = idx->d_klo; v.khi = idx->d_khi; v.focc = idx->d_focc; v.rmq = idx->d_rmq;
             v.n = (uint32_t)idx->len; v.levels = idx->rmq_levels; v.first_bad = idx->first_bad; }
  return v;
}

const uint8_t* pgpu_index_genomic(const pgpu_index* idx) { return idx->d_gen; }
size_t pgpu_index_length(const pgpu_index* idx) { return idx->len; }

namespace {

// ---------------------------------------------------------------------------------------------
// suffix array by prefix doubling
// ---------------------------------------------------------------------------------------------
__global__ void sa_init_kernel(const uint8_t* __restrict__ T, uint32_t n, uint32_t* rank, uint32_t* sa) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { rank[i] = (uint32_t)T[i] + 1u; sa[i] = i; }
}

__global__ void sa_keys_kernel(const uint32_t* __restrict__ rank, const uint32_t* __restrict__ sa,
                               uint32_t n, uint32_t h, unsigned long long* keys) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = sa[k];
  const uint32_t r2 = (i + h < n) ? rank[i + h] : 0u;          // past the end sorts first ('$')
  keys[k] = ((unsigned long long)rank[i] << 32) | r2;
}

__global__ void sa_flags_kernel(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t* flags) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1u : 0u;
}

__global__ void sa_rerank_kernel(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ newrank_sorted,
                                 uint32_t n, uint32_t* rank) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) rank[sa[k]] = newrank_sorted[k];
}

// LCP of neighbouring suffixes from the rank arrays the prefix doubling went through: ranks[r] orders
// the suffixes by their first 2^r characters (the end of the text sorting first), so two suffixes
// share 2^r more characters exactly when their ranks[r] agree; descending r adds up the exact LCP
// in ~log n steps per pair, however repetitive the sequence is (a character-by-character scan is
// quadratic on a long exact repeat).
__global__ void lcp_from_ranks_kernel(const uint32_t* const* __restrict__ ranks, int n_rounds,
                                      const uint32_t* __restrict__ sa, uint32_t n, uint32_t* __restrict__ lcp) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n) return;
  if (k == 0 || k == n) { lcp[k] = 0; return; }
  const uint32_t a = sa[k - 1], b = sa[k];
  uint32_t l = 0;
  for (int r = n_rounds - 1; r >= 0; --r) {
    if (a + l < n && b + l < n && ranks[r][a + l] == ranks[r][b + l]) l += 1u << r;
  }
  lcp[k] = l;
}

__device__ __forceinline__ int kmer_base(uint32_t c) {
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}
// 2-bit code of s[0..KTAB), or -1 when a character is not A/C/G/T (`avail` readable characters)
__device__ __forceinline__ int kmer_code(const uint8_t* __restrict__ s, uint32_t avail) {
  if (avail < KTAB) return -1;
  int code = 0;
#pragma unroll
  for (uint32_t x = 0; x < KTAB; ++x) {
    const int b = kmer_base(s[x]);
    if (b < 0) return -1;
    code = (code << 2) | b;
  }
  return code;
}
// suffixes sharing a KTAB-mer are contiguous in the suffix array: mark where each run starts/ends
__global__ void kmer_table_kernel(const uint8_t* __restrict__ T, uint32_t n, const uint32_t* __restrict__ sa,
                                  uint32_t* __restrict__ klo, uint32_t* __restrict__ khi) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int c = kmer_code(T + sa[k], n - sa[k]);
  if (c < 0) return;
  const int prev = k > 0 ? kmer_code(T + sa[k - 1], n - sa[k - 1]) : -1;
  const int next = k + 1 < n ? kmer_code(T + sa[k + 1], n - sa[k + 1]) : -1;
  if (prev != c) klo[c] = k;
  if (next != c) khi[c] = k + 1;
}

// first occurrence of every l-mer (l = 1..8) that starts at t: tables zero-based at (4^l - 4) / 3
__global__ void first_occ_kernel(const uint8_t* __restrict__ T, uint32_t n, uint32_t* __restrict__ focc) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint32_t code = 0, off = 0, width = 4;
  for (uint32_t l = 1; l <= 8 && t + l <= n; ++l) {
    const int b = kmer_base(T[t + l - 1]);
    if (b < 0) break;
    code = code * 4 + (uint32_t)b;
    // (a first occurrence lies early in the sequence: nearly every thread sees a smaller value already there and
    // skips the atomic -- a million atomics on the four entries of l = 1 took 8 ms on a 1 Mb sequence)
    if (t < focc[off + code]) atomicMin(&focc[off + code], t);
    off += width; width *= 4;
  }
}
// one level of the sparse table: out[k] = min(in[k], in[k + half]) for k + 2 * half <= n
__global__ void rmq_level_kernel(const uint32_t* __restrict__ in, uint32_t n, uint32_t half, uint32_t* __restrict__ out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k + 2 * half <= n) out[k] = min(in[k], in[k + half]);
}

// ---------------------------------------------------------------------------------------------
// pairings
// ---------------------------------------------------------------------------------------------
constexpr uint32_t NONE = 0xFFFFFFFFu;

struct PairParams {
  uint32_t L;
  double rate;
};

// Sequences at 2 bits per base.  Both the genomic sequence and the patterns are compared sixteen bases per
// 32-bit word (one XOR and a count of trailing zeros) instead of a byte per load: a maximal pairing is as long
// as an exon, and pair_locate / pair_count / pair_fill walk it once per occurrence.  A flag bit per base marks
// what is not an upper-case A, C, G or T (N, the '*' and '#' that mask poly-A/T stretches, lower case): a
// window that holds a flagged base on either side is compared on the bytes, so the result is byte equality
// everywhere (an N equals an N, src/max-emb-graph.c compares characters).
struct PackedSeq { const uint32_t* __restrict__ code; const uint32_t* __restrict__ bad; };

__global__ void pack2_kernel(const uint8_t* __restrict__ src, unsigned long long n, uint32_t* __restrict__ code,
                             uint32_t* __restrict__ bad, unsigned long long n_bad_words) {
  const unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;    // one flag word = 32 bases
  if (w >= n_bad_words) return;
  uint32_t c0 = 0, c1 = 0, b = 0;
  for (uint32_t k = 0; k < 32; ++k) {
    const unsigned long long p = w * 32 + k;
    const int x = p < n ? kmer_base(src[p]) : -1;
    if (x < 0) b |= 1u << k;
    else if (k < 16) c0 |= (uint32_t)x << (2 * k);
    else c1 |= (uint32_t)x << (2 * (k - 16));
  }
  code[2 * w] = c0; code[2 * w + 1] = c1; bad[w] = b;
}

__device__ __forceinline__ uint32_t pk_win16(const uint32_t* __restrict__ code, unsigned long long pos) {
  const unsigned long long w = pos >> 4;
  const unsigned long long x = ((unsigned long long)code[w + 1] << 32) | code[w];
  return (uint32_t)(x >> (2u * (uint32_t)(pos & 15u)));
}
__device__ __forceinline__ uint32_t pk_bad16(const uint32_t* __restrict__ bad, unsigned long long pos) {
  const unsigned long long w = pos >> 5;
  const unsigned long long x = ((unsigned long long)bad[w + 1] << 32) | bad[w];
  return (uint32_t)(x >> (uint32_t)(pos & 31u)) & 0xFFFFu;
}
// number of leading characters on which A[a..] and B[b..] agree, among the first `rem`
__device__ __forceinline__ uint32_t pk_match(const uint8_t* __restrict__ A, const PackedSeq Ap, unsigned long long a,
                                             const uint8_t* __restrict__ B, const PackedSeq Bp, unsigned long long b,
                                             uint32_t rem) {
  uint32_t l = 0;
  while (l < rem) {
    const uint32_t k = rem - l < 16u ? rem - l : 16u;
    const uint32_t mask = k < 16u ? (1u << k) - 1u : 0xFFFFu;
    if ((pk_bad16(Ap.bad, a + l) | pk_bad16(Bp.bad, b + l)) & mask) {
      uint32_t q = 0;
      while (q < k && A[a + l + q] == B[b + l + q]) ++q;
      l += q;
      if (q < k) break;
      continue;
    }
    uint32_t x = pk_win16(Ap.code, a + l) ^ pk_win16(Bp.code, b + l);
    if (k < 16u) x &= (1u << (2u * k)) - 1u;
    if (x) { l += (uint32_t)__builtin_ctz(x) >> 1; break; }
    l += k;
  }
  return l;
}

// what the comparisons of one pattern need: the genomic sequence and the pattern blob, as bytes and packed
struct CmpCtx {
  const uint8_t* __restrict__ T; PackedSeq Tp; uint32_t n;
  const uint8_t* __restrict__ pats; PackedSeq Pp;
};

// compare suffix t (from character `skip` on) with q[skip..d), q = pats + qabs: -1 / 0 (q[0..d) is a prefix) / +1.
// The caller guarantees the first `skip` characters are equal.
__device__ __forceinline__ int cmp_suffix(const CmpCtx& cx, uint32_t t, unsigned long long qabs, uint32_t d, uint32_t skip) {
  if (skip >= d) return 0;
  if (t + skip >= cx.n) return -1;                  // the suffix ended: it sorts first
  const uint32_t avail = cx.n - t - skip, want = d - skip;
  const uint32_t k = pk_match(cx.T, cx.Tp, (unsigned long long)t + skip, cx.pats, cx.Pp, qabs + skip, avail < want ? avail : want);
  if (k == want) return 0;
  if (k == avail) return -1;
  const uint32_t c = cx.T[t + skip + k], e = cx.pats[qabs + skip + k];
  return c < e ? -1 : 1;
}

// [lo,hi) of suffixes in sa[from,to) having q[0..d) as a prefix (all share q[0..skip))
__device__ void sa_interval(const CmpCtx& cx, const uint32_t* __restrict__ sa,
                            unsigned long long qabs, uint32_t d, uint32_t skip,
                            uint32_t from, uint32_t to, uint32_t* lo, uint32_t* hi) {
  uint32_t a = from, b = to;
  while (a < b) {
    const uint32_t mid = a + ((b - a) >> 1);
    if (cmp_suffix(cx, sa[mid], qabs, d, skip) < 0) a = mid + 1; else b = mid;
  }
  *lo = a;
  b = to;
  while (a < b) {
    const uint32_t mid = a + ((b - a) >> 1);
    if (cmp_suffix(cx, sa[mid], qabs, d, skip) <= 0) a = mid + 1; else b = mid;
  }
  *hi = a;
}

__device__ __forceinline__ bool prev_excluded(const uint8_t* __restrict__ T, uint32_t t,
                                              const uint8_t* __restrict__ P, uint32_t i) {
  return i > 0 && t > 0 && T[t - 1] == P[i - 1];     // src/max-emb-graph.c:178-181,195
}

// length of the match of P[i..m) with T[t..), known to be at least `from` (P = pats + pbase)
__device__ __forceinline__ uint32_t extend(const CmpCtx& cx, uint32_t t, unsigned long long pbase, uint32_t m, uint32_t i,
                                           uint32_t from) {
  if (i + from >= m || t + from >= cx.n) return from;
  const uint32_t rp = m - i - from, rt = cx.n - t - from;
  return from + pk_match(cx.T, cx.Tp, (unsigned long long)t + from, cx.pats, cx.Pp, pbase + i + from, rp < rt ? rp : rt);
}

// How many times the reference lists the occurrence t == 0 of position i (oracle/pairing_oracle.c
// states and tests the rule; src/max-emb-graph.c:168-216, src/aug_suffix_tree.c:183-192): suffix 0
// has no preceding character, sits in the slice of every symbol, and is reported once per slice walked
// -- through the unguarded loop where the child already reported holds an occurrence preceded by
// that symbol, through the guarded one (key 0, or key 1 next to the preceding symbol's key 0)
// elsewhere.  *l0 = lcp(P[i..], T[0..]); the result only matters when l0 reaches the threshold.
__device__ uint32_t zero_copies(const CmpCtx& cx, unsigned long long pbase, const uint32_t* __restrict__ sa,
                                const uint8_t* __restrict__ key, uint32_t sigma, const uint8_t* __restrict__ P,
                                uint32_t m, uint32_t i, uint32_t L, uint32_t lo, uint32_t hi, uint32_t l0) {
  const uint8_t* __restrict__ T = cx.T;
  unsigned long long present[4] = {0, 0, 0, 0};
  for (uint32_t k = lo; k < hi; ++k) {
    const uint32_t t = sa[k];
    if (t == 0 || prev_excluded(T, t, P, i)) continue;
    if (extend(cx, t, pbase, m, i, L) > l0) { const uint32_t q = key[T[t - 1]]; present[q >> 6] |= 1ull << (q & 63u); }
  }
  const uint32_t sk = i > 0 ? key[P[i - 1]] : sigma;
  uint32_t copies = 0;
  for (uint32_t k = 0; k < sigma; ++k) {
    if (k == sk) continue;
    if ((present[k >> 6] >> (k & 63u)) & 1ull) ++copies;
    else if (k == 0 || (k == 1 && sk == 0)) ++copies;
  }
  return copies;
}

// one wave per pattern, lanes stride over its positions
__global__ __launch_bounds__(64)
void pair_locate_kernel(const uint8_t* __restrict__ T, uint32_t n, const uint32_t* __restrict__ sa,
                        const uint32_t* __restrict__ klo, const uint32_t* __restrict__ khi,
                        const uint8_t* __restrict__ pats, const unsigned long long* __restrict__ pat_off,
                        PairParams prm, uint32_t* __restrict__ lo_out, uint32_t* __restrict__ hi_out,
                        uint32_t* __restrict__ a_out, const PackedSeq Tp, const PackedSeq Pp) {
  const CmpCtx cx{T, Tp, n, pats, Pp};
  const unsigned long long base = pat_off[blockIdx.x];
  const uint32_t m = (uint32_t)(pat_off[blockIdx.x + 1] - base);
  const uint8_t* P = pats + base;
  for (uint32_t i = threadIdx.x; i < m; i += 64) {
    uint32_t lo = 0, hi = 0, A = 0;
    if (m - i >= prm.L) {
      const int code = prm.L >= KTAB ? kmer_code(P + i, m - i) : -1;
      if (code >= 0) {                              // the k-mer's run, then bisect on characters KTAB..L
        const uint32_t from = klo[code], to = khi[code];
        if (from < to) sa_interval(cx, sa, base + i, prm.L, KTAB, from, to, &lo, &hi);
      } else {
        sa_interval(cx, sa, base + i, prm.L, 0, 0, n, &lo, &hi);
      }
      for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t t = sa[k];
        if (prev_excluded(T, t, P, i)) continue;
        const uint32_t l = extend(cx, t, base, m, i, prm.L);
        A = l > A ? l : A;
      }
    }
    lo_out[base + i] = lo; hi_out[base + i] = hi; a_out[base + i] = A;
  }
}

// one thread per pattern: the locus-depth recurrence (sequential in i)
__global__ __launch_bounds__(64)
void pair_chain_kernel(const uint8_t* __restrict__ T, uint32_t n, const uint32_t* __restrict__ sa,
                       const uint32_t* __restrict__ lcp, const uint8_t* __restrict__ pats,
                       const unsigned long long* __restrict__ pat_off, uint32_t n_pat, PairParams prm,
                       const uint32_t* __restrict__ lo_in, const uint32_t* __restrict__ hi_in,
                       const uint32_t* __restrict__ a_in, uint32_t* __restrict__ thr_out, const PackedSeq Tp, const PackedSeq Pp) {
  const CmpCtx cx{T, Tp, n, pats, Pp};
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pat) return;
  const unsigned long long base = pat_off[p];
  const uint32_t m = (uint32_t)(pat_off[p + 1] - base);
  const uint8_t* P = pats + base;
  uint32_t s = 0;
  for (uint32_t i = 0; i < m; ++i) {
    const uint32_t A = a_in[base + i];
    const uint32_t D = A > s ? A : s;
    if (D < prm.L) { thr_out[base + i] = NONE; s = 0; continue; }
    const double scaled = (double)D * prm.rate;                  // src/max-emb-graph.c:273-274
    const double thr_d = scaled > (double)prm.L ? scaled : (double)prm.L;
    thr_out[base + i] = (uint32_t)thr_d;
    const uint32_t lo = lo_in[base + i], hi = hi_in[base + i];
    uint32_t l2 = lo, h2 = hi;
    if (hi - lo > 1 && D > prm.L) sa_interval(cx, sa, base + i, D, prm.L, lo, hi, &l2, &h2);
    const uint32_t pl = lcp[l2], ph = lcp[h2];
    const uint32_t par = pl > ph ? pl : ph;
    if (par == 0) { s = 0; continue; }
    bool at_node = false;
    if (h2 - l2 >= 2) {
      const uint32_t t1 = sa[l2], t2 = sa[h2 - 1];
      const int c1 = t1 + D < n ? (int)T[t1 + D] : -1;
      const int c2 = t2 + D < n ? (int)T[t2 + D] : -1;
      at_node = c1 != c2;
    }
    s = at_node ? D - 1 : par - 1;
  }
}

// number of occurrences at or above the threshold (before the filters)
__global__ __launch_bounds__(64)
void pair_count_kernel(const uint8_t* __restrict__ T, uint32_t n, const uint32_t* __restrict__ sa,
                       const uint8_t* __restrict__ key, uint32_t sigma,
                       const uint8_t* __restrict__ pats, const unsigned long long* __restrict__ pat_off,
                       PairParams prm, const uint32_t* __restrict__ lo_in, const uint32_t* __restrict__ hi_in,
                       const uint32_t* __restrict__ thr_in, uint32_t* __restrict__ cnt_out, const PackedSeq Tp, const PackedSeq Pp) {
  const CmpCtx cx{T, Tp, n, pats, Pp};
  const unsigned long long base = pat_off[blockIdx.x];
  const uint32_t m = (uint32_t)(pat_off[blockIdx.x + 1] - base);
  const uint8_t* P = pats + base;
  for (uint32_t i = threadIdx.x; i < m; i += 64) {
    const uint32_t thr = thr_in[base + i];
    uint32_t c = 0;
    if (thr != NONE) {
      const uint32_t lo = lo_in[base + i], hi = hi_in[base + i];
      for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t t = sa[k];
        if (prev_excluded(T, t, P, i)) continue;
        const uint32_t l = extend(cx, t, base, m, i, prm.L);
        if (l < thr) continue;
        c += t == 0 ? zero_copies(cx, base, sa, key, sigma, P, m, i, prm.L, lo, hi, l) : 1u;
      }
    }
    cnt_out[base + i] = c;
  }
}

struct Cand { uint32_t t, l; };

// write the candidates of each position, sort them by t, apply filter (a)
// (src/max-emb-graph.c:301-334) and move the survivors to the front of the position's slot
__global__ __launch_bounds__(64)
void pair_fill_kernel(const uint8_t* __restrict__ T, uint32_t n, const uint32_t* __restrict__ sa,
                      const uint8_t* __restrict__ key, uint32_t sigma,
                      const uint8_t* __restrict__ pats, const unsigned long long* __restrict__ pat_off,
                      PairParams prm, const uint32_t* __restrict__ lo_in, const uint32_t* __restrict__ hi_in,
                      const uint32_t* __restrict__ thr_in, const unsigned long long* __restrict__ cand_off,
                      Cand* __restrict__ cand, uint32_t* __restrict__ cnt_a, const PackedSeq Tp, const PackedSeq Pp) {
  const CmpCtx cx{T, Tp, n, pats, Pp};
  const unsigned long long base = pat_off[blockIdx.x];
  const uint32_t m = (uint32_t)(pat_off[blockIdx.x + 1] - base);
  const uint8_t* P = pats + base;
  for (uint32_t i = threadIdx.x; i < m; i += 64) {
    const uint32_t thr = thr_in[base + i];
    uint32_t c = 0;
    if (thr != NONE) {
      Cand* slot = cand + cand_off[base + i];
      const uint32_t lo = lo_in[base + i], hi = hi_in[base + i];
      for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t t = sa[k];
        if (prev_excluded(T, t, P, i)) continue;
        const uint32_t l = extend(cx, t, base, m, i, prm.L);
        if (l < thr) continue;
        // insertion sort by t (distinct, except the copies of t == 0, which stay together in front)
        const uint32_t copies = t == 0 ? zero_copies(cx, base, sa, key, sigma, P, m, i, prm.L, lo, hi, l) : 1u;
        for (uint32_t cpy = 0; cpy < copies; ++cpy) {
          uint32_t q = c++;
          while (q > 0 && slot[q - 1].t > t) { slot[q] = slot[q - 1]; --q; }
          slot[q].t = t; slot[q].l = l;
        }
      }
      // filter (a): PJ dies when an earlier PI (smaller t) covers it or is its twin shifted by one.
      // The reference tests against ALL earlier entries, dead ones included, so deaths are
      // decided on the unmodified sorted list: mark with the top bit of l, then compact.
      for (uint32_t j = c; j-- > 1;) {
        const uint32_t tj = slot[j].t, lj = slot[j].l & 0x7FFFFFFFu;
        for (uint32_t q = j; q-- > 0;) {
          const uint32_t ti = slot[q].t, li = slot[q].l & 0x7FFFFFFFu;
          if ((tj > ti && tj + lj <= ti + li) || (tj == ti + 1 && lj == li)) { slot[j].l |= 0x80000000u; break; }
        }
      }
      uint32_t w = 0;
      for (uint32_t k = 0; k < c; ++k)
        if (!(slot[k].l & 0x80000000u)) slot[w++] = slot[k];
      c = w;
    }
    cnt_a[base + i] = c;
  }
}

// filter (b) (src/max-emb-graph.c:349-375): position i loses I1 when position i-1 (after filter
// (a), before (b)) holds I with I.t == I1.t and I.l >= I1.l.  Flags only: lists stay intact.
__global__ __launch_bounds__(64)
void pair_cross_kernel(const unsigned long long* __restrict__ pat_off,
                       const unsigned long long* __restrict__ cand_off, const Cand* __restrict__ cand,
                       const uint32_t* __restrict__ cnt_a, uint8_t* __restrict__ keep,
                       uint32_t* __restrict__ cnt_b) {
  const unsigned long long base = pat_off[blockIdx.x];
  const uint32_t m = (uint32_t)(pat_off[blockIdx.x + 1] - base);
  for (uint32_t i = threadIdx.x; i < m; i += 64) {
    const uint32_t c = cnt_a[base + i];
    const unsigned long long off = cand_off[base + i];
    uint32_t kept = 0;
    if (c) {
      const uint32_t cp = i > 0 ? cnt_a[base + i - 1] : 0;
      const Cand* prev = i > 0 ? cand + cand_off[base + i - 1] : nullptr;
      for (uint32_t k = 0; k < c; ++k) {
        const Cand x = cand[off + k];
        bool rim = false;
        for (uint32_t q = 0; q < cp && !rim; ++q) rim = prev[q].t == x.t && prev[q].l >= x.l;
        keep[off + k] = rim ? 0 : 1;
        kept += rim ? 0 : 1;
      }
    }
    cnt_b[base + i] = kept;
  }
}

__global__ __launch_bounds__(64)
void pair_emit_kernel(const unsigned long long* __restrict__ pat_off,
                      const unsigned long long* __restrict__ cand_off, const Cand* __restrict__ cand,
                      const uint32_t* __restrict__ cnt_a, const uint8_t* __restrict__ keep,
                      const unsigned long long* __restrict__ out_off, pgpu_pairing* __restrict__ out,
                      unsigned long long* __restrict__ out_first, uint32_t n_pat, unsigned long long total_pos) {
  const unsigned long long base = pat_off[blockIdx.x];
  const uint32_t m = (uint32_t)(pat_off[blockIdx.x + 1] - base);
  if (threadIdx.x == 0) {
    out_first[blockIdx.x] = base < total_pos ? out_off[base] : out_off[total_pos];
    if (blockIdx.x == n_pat - 1) out_first[n_pat] = out_off[total_pos];
  }
  for (uint32_t i = threadIdx.x; i < m; i += 64) {
    const uint32_t c = cnt_a[base + i];
    const unsigned long long off = cand_off[base + i];
    unsigned long long w = out_off[base + i];
    for (uint32_t k = 0; k < c; ++k) {
      if (!keep[off + k]) continue;
      out[w].p = (int32_t)i; out[w].t = (int32_t)cand[off + k].t; out[w].l = (int32_t)cand[off + k].l;
      ++w;
    }
  }
}

template <class T> hipError_t dmalloc(T** p, size_t count) {
  return hipMalloc((void**)p, (count ? count : 1) * sizeof(T));
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C-ABI: index
// ---------------------------------------------------------------------------------------------
// alphabet keys of the genomic sequence (host), uploaded with the index
static hipError_t upload_keys(pgpu_index* idx, const char* genomic, size_t len, hipStream_t st) {
  bool seen[256] = {false};
  for (size_t i = 0; i < len; ++i) seen[(unsigned char)genomic[i]] = true;
  uint8_t key[256];
  uint32_t sigma = 0;
  for (int c = 0; c < 256; ++c) if (seen[c]) key[c] = (uint8_t)sigma++;
  for (int c = 0; c < 256; ++c) if (!seen[c]) key[c] = (uint8_t)(sigma > 255 ? 255 : sigma);
  idx->sigma = sigma;
  hipError_t e = hipMalloc((void**)&idx->d_key, 256);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(idx->d_key, key, 256, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(st);           // `key` is a local
}

// src[0..n) -> code (2 * ceil(n / 32) words, + 4 of slack the windows may read) and flag words
static hipError_t pack_sequence(const uint8_t* src, size_t n, uint32_t* code, uint32_t* bad, hipStream_t st) {
  const size_t bad_words = (n + 31) / 32;
  hipError_t e = hipMemsetAsync(code + 2 * bad_words, 0, 4 * sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(bad + bad_words, 0xFF, 4 * sizeof(uint32_t), st);
  if (e != hipSuccess || bad_words == 0) return e;
  hipLaunchKernelGGL(pack2_kernel, dim3((unsigned)((bad_words + 255) / 256)), dim3(256), 0, st, src, (unsigned long long)n, code, bad,
                     (unsigned long long)bad_words);
  return hipGetLastError();
}

// tables derived from (sequence, suffix array): see LcfIndexView
static hipError_t build_lcf_tables(pgpu_index* idx, const char* genomic, hipStream_t st) {
  const uint32_t n = (uint32_t)idx->len;
  {
    const size_t bad_words = ((size_t)n + 31) / 32;
    hipError_t pe = hipMalloc((void**)&idx->d_code, (2 * bad_words + 4) * sizeof(uint32_t));
    if (pe != hipSuccess) return pe;
    pe = hipMalloc((void**)&idx->d_bad, (bad_words + 4) * sizeof(uint32_t));
    if (pe != hipSuccess) return pe;
    pe = pack_sequence(idx->d_gen, n, idx->d_code, idx->d_bad, st);
    if (pe != hipSuccess) return pe;
  }
  uint32_t fb = n;
  for (uint32_t i = 0; i < n; ++i) { const char c = genomic[i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') { fb = i; break; } }
  idx->first_bad = fb;
  // The tables of the suffix-array LCF are optional: floor(log2 n) x n x 4 B of range minima is 14 MB for a 200 kb
  // gene but 10 GB for 100 Mb.  Without them (switched off, a sequence beyond PGPU_LCF_SA_MAX_BASES -- default
  // 2^26 --, or no memory) every LCF job takes lcf_kernel: pgpu_dp_plan_create_parts looks at d_focc / d_rmq.
  {
    const char* sw = getenv("PGPU_LCF_SA");
    const char* mx = getenv("PGPU_LCF_SA_MAX_BASES");
    const unsigned long long max_bases = mx && atoll(mx) > 0 ? (unsigned long long)atoll(mx) : (1ull << 26);
    if ((sw && sw[0] == '0' && sw[1] == '\0') || n > max_bases) return hipSuccess;
  }
  hipError_t e = hipMalloc((void**)&idx->d_focc, LCF_FOCC_ENTRIES * sizeof(uint32_t));
  if (e != hipSuccess) { idx->d_focc = nullptr; (void)hipGetLastError(); return hipSuccess; }
  e = hipMemsetAsync(idx->d_focc, 0xFF, LCF_FOCC_ENTRIES * sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  uint32_t levels = 0;
  while (n >= 2 && (2u << levels) <= n) ++levels;              // levels j = 1..levels with 2^j <= n
  idx->rmq_levels = levels;
  e = hipMalloc((void**)&idx->d_rmq, ((size_t)(levels ? levels : 1) * (n ? n : 1)) * sizeof(uint32_t));
  if (e != hipSuccess) {                                       // no room for the range minima: the matrix kernel serves
    idx->d_rmq = nullptr; idx->rmq_levels = 0; (void)hipGetLastError();
    hipFree(idx->d_focc); idx->d_focc = nullptr;
    return hipSuccess;
  }
  if (n == 0) return hipSuccess;
  const dim3 blk(256), grd((n + 255) / 256);
  hipLaunchKernelGGL(first_occ_kernel, grd, blk, 0, st, idx->d_gen, n, idx->d_focc);
  for (uint32_t j = 1; j <= levels; ++j)
    hipLaunchKernelGGL(rmq_level_kernel, grd, blk, 0, st, j == 1 ? idx->d_sa : idx->d_rmq + (size_t)(j - 2) * n, n, 1u << (j - 1),
                       idx->d_rmq + (size_t)(j - 1) * n);
  return hipGetLastError();
}

#define TRY_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) { rc = pgpu_ctx_fail(ctx, e_ == hipErrorOutOfMemory ? PGPU_ENOMEM : PGPU_EDEVICE, \
                                               hipGetErrorString(e_)); goto done; }          \
  } while (0)

static double idx_now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

extern "C" int pgpu_index_build(pgpu_ctx* ctx, const char* genomic, size_t len, pgpu_index** out) {
  if (!ctx || !out || (len && !genomic)) return PGPU_EINVAL;
  const bool timing = getenv("PGPU_INDEX_TIMING") != nullptr;      // phases of the construction, to stderr
  double t_mark = idx_now_ms();
  int n_rounds = 0;
  auto mark = [&](const char* what, hipStream_t s) {
    if (!timing) return;
    hipStreamSynchronize(s);
    const double t = idx_now_ms();
    fprintf(stderr, "* index build: %-28s %7.2f ms\n", what, t - t_mark);
    t_mark = t;
  };
  *out = nullptr;
  if (len >= (1u << 28)) return pgpu_ctx_fail(ctx, PGPU_ERANGE, "genomic longer than 2^28");
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  pgpu_index* idx = new (std::nothrow) pgpu_index();
  if (!idx) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of host memory");
  idx->len = len;
  hipStream_t st = pgpu_ctx_stream(ctx);
  const uint32_t n = (uint32_t)len;
  int rc = PGPU_OK;
  uint32_t *rank = nullptr, *sa2 = nullptr, *flags = nullptr, *newrank = nullptr;
  std::vector<uint32_t*> round_ranks;        // rank arrays by 2^r-character prefixes, r = 0, 1, ...
  const uint32_t** d_round_ptrs = nullptr;
  unsigned long long *keys = nullptr, *keys2 = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  const dim3 blk(256), grd((n + 255) / 256 + 1);

  TRY_HIP(hipMalloc((void**)&idx->d_gen, len + 64));
  TRY_HIP(hipMemsetAsync(idx->d_gen, 0, len + 64, st));
  if (len) TRY_HIP(hipMemcpyAsync(idx->d_gen, genomic, len, hipMemcpyHostToDevice, st));
  TRY_HIP(upload_keys(idx, genomic, len, st));
  TRY_HIP(dmalloc(&idx->d_sa, n + 1));
  TRY_HIP(dmalloc(&idx->d_lcp, n + 2));
  if (n > 0) {
    TRY_HIP(dmalloc(&rank, n)); TRY_HIP(dmalloc(&sa2, n)); TRY_HIP(dmalloc(&flags, n));
    TRY_HIP(dmalloc(&newrank, n)); TRY_HIP(dmalloc(&keys, n)); TRY_HIP(dmalloc(&keys2, n));
    {
      size_t b1 = 0, b2 = 0;
      TRY_HIP(rocprim::radix_sort_pairs(nullptr, b1, keys, keys2, idx->d_sa, sa2, n, 0, 64, st));
      TRY_HIP(rocprim::inclusive_scan(nullptr, b2, flags, newrank, n, rocprim::plus<uint32_t>(), st));
      tmp_bytes = b1 > b2 ? b1 : b2;
      TRY_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    }
    hipLaunchKernelGGL(sa_init_kernel, grd, blk, 0, st, idx->d_gen, n, rank, idx->d_sa);
    mark("allocations + upload", st);
    for (uint32_t h = 0;; h = h ? h * 2 : 1) {
      ++n_rounds;
      // h == 0: sort by the first character alone (key low half = 0 because i+0 < n gives rank[i]
      // again -- harmless duplicate), afterwards by (rank of 2^k prefix, rank of the next 2^k)
      hipLaunchKernelGGL(sa_keys_kernel, grd, blk, 0, st, rank, idx->d_sa, n, h ? h : n, keys);
      size_t b = tmp_bytes;
      TRY_HIP(rocprim::radix_sort_pairs(tmp, b, keys, keys2, idx->d_sa, sa2, n, 0, 64, st));
      hipLaunchKernelGGL(sa_flags_kernel, grd, blk, 0, st, keys2, n, flags);
      b = tmp_bytes;
      TRY_HIP(rocprim::inclusive_scan(tmp, b, flags, newrank, n, rocprim::plus<uint32_t>(), st));
      hipLaunchKernelGGL(sa_rerank_kernel, grd, blk, 0, st, sa2, newrank, n, rank);
      {                                        // this round ranked by the first max(1, 2h) characters
        uint32_t* keep = nullptr;
        TRY_HIP(dmalloc(&keep, n));
        round_ranks.push_back(keep);
        TRY_HIP(hipMemcpyAsync(keep, rank, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
      }
      TRY_HIP(hipMemcpyAsync(idx->d_sa, sa2, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
      uint32_t distinct = 0;
      TRY_HIP(hipMemcpyAsync(&distinct, newrank + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      TRY_HIP(hipStreamSynchronize(st));
      if (distinct == n) break;
      if (h >= n) { rc = pgpu_ctx_fail(ctx, PGPU_EDEVICE, "suffix array construction did not converge"); goto done; }
    }
  }
  if (timing) fprintf(stderr, "* index build: %d sort rounds\n", n_rounds);
  mark("suffix array (sort rounds)", st);
  {
    // round 0 ranked by 1 character (h == 0), round j >= 1 by 2^j characters: ranks[r] <-> 2^r
    const int nr = (int)round_ranks.size();
    TRY_HIP(hipMalloc((void**)&d_round_ptrs, (nr ? nr : 1) * sizeof(uint32_t*)));
    if (nr) TRY_HIP(hipMemcpyAsync(d_round_ptrs, round_ranks.data(), nr * sizeof(uint32_t*), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lcp_from_ranks_kernel, grd, blk, 0, st, (const uint32_t* const*)d_round_ptrs, nr, idx->d_sa, n, idx->d_lcp);
  }
  mark("LCP from the ranks", st);
  TRY_HIP(dmalloc(&idx->d_klo, KTAB_ENTRIES));
  TRY_HIP(dmalloc(&idx->d_khi, KTAB_ENTRIES));
  TRY_HIP(hipMemsetAsync(idx->d_klo, 0, KTAB_ENTRIES * sizeof(uint32_t), st));
  TRY_HIP(hipMemsetAsync(idx->d_khi, 0, KTAB_ENTRIES * sizeof(uint32_t), st));
  if (n > 0) hipLaunchKernelGGL(kmer_table_kernel, grd, blk, 0, st, idx->d_gen, n, idx->d_sa, idx->d_klo, idx->d_khi);
  mark("8-mer table", st);
  TRY_HIP(build_lcf_tables(idx, genomic, st));
  TRY_HIP(hipStreamSynchronize(st));
  mark("packing + LCF tables", st);
  TRY_HIP(hipGetLastError());
done:
  hipFree(rank); hipFree(sa2); hipFree(flags); hipFree(newrank); hipFree(keys); hipFree(keys2); hipFree(tmp);
  for (uint32_t* q : round_ranks) hipFree(q);
  hipFree(d_round_ptrs);
  if (rc != PGPU_OK) { hipFree(idx->d_gen); hipFree(idx->d_sa); hipFree(idx->d_lcp); hipFree(idx->d_klo); hipFree(idx->d_khi); hipFree(idx->d_key); hipFree(idx->d_focc); hipFree(idx->d_rmq); hipFree(idx->d_code); hipFree(idx->d_bad); delete idx; return rc; }
  *out = idx;
  return PGPU_OK;
}

// ---- the index on disk: a gene is usually processed many times (parameter studies, re-runs of the
// pipeline), the index depends on the genomic sequence alone ------------------------------------
namespace {
// version 2: + checksum of the payload (a file whose header is right and whose tables are not -- torn by
// concurrent writers, truncated and padded, bit rot -- must not reach the device: an out-of-range
// suffix-array entry is an out-of-bounds read in every pairing kernel)
struct IndexFileHeader { char magic[8]; uint32_t version, ktab; uint64_t len, hash, payload_hash; };
const char INDEX_MAGIC[8] = { 'P', 'G', 'P', 'U', 'I', 'D', 'X', '1' };
constexpr uint32_t INDEX_VERSION = 2;
// word-wise FNV-style mix of the payload (1 Mb genomic = 9 MB of tables: a byte-wise loop would cost
// more than rebuilding the index)
uint64_t words_hash(const uint32_t* w, size_t n) {
  uint64_t h0 = 1469598103934665603ull, h1 = 0x9e3779b97f4a7c15ull;
  size_t i = 0;
  for (; i + 1 < n; i += 2) { h0 = (h0 ^ w[i]) * 1099511628211ull; h1 = (h1 ^ w[i + 1]) * 0x100000001b3ull; }
  if (i < n) h0 = (h0 ^ w[i]) * 1099511628211ull;
  return h0 ^ (h1 * 0xff51afd7ed558ccdull) ^ (uint64_t)n;
}
uint64_t fnv1a64(const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
}  // namespace

extern "C" int pgpu_index_save(pgpu_ctx* ctx, const pgpu_index* idx, const char* genomic, const char* path) {
  if (!ctx || !idx || !path || (idx->len && !genomic)) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  const size_t n = idx->len;
  std::vector<uint32_t> host(2 * n + 1 + 2 * (size_t)KTAB_ENTRIES);
  hipStream_t st = pgpu_ctx_stream(ctx);
  if ((n && hipMemcpyAsync(host.data(), idx->d_sa, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess) ||
      hipMemcpyAsync(host.data() + n, idx->d_lcp, (n + 1) * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(host.data() + 2 * n + 1, idx->d_klo, KTAB_ENTRIES * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(host.data() + 2 * n + 1 + KTAB_ENTRIES, idx->d_khi, KTAB_ENTRIES * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return pgpu_ctx_fail(ctx, PGPU_EDEVICE, "index download failed");
  IndexFileHeader h;
  memcpy(h.magic, INDEX_MAGIC, 8); h.version = INDEX_VERSION; h.ktab = KTAB; h.len = n; h.hash = fnv1a64(genomic, n);
  h.payload_hash = words_hash(host.data(), host.size());
  // written under a temporary name of its OWN (several ranks, or several est-fact processes on the same
  // gene, may miss the cache and save at the same time) and renamed: a reader sees a whole file or none
  char tmpl[4200];
  if (snprintf(tmpl, sizeof tmpl, "%s.tmp.XXXXXX", path) >= (int)sizeof tmpl) return pgpu_ctx_fail(ctx, PGPU_EINVAL, "index path too long");
  const int fd = mkstemp(tmpl);
  if (fd < 0) return pgpu_ctx_fail(ctx, PGPU_EINVAL, "cannot create the index file");
  (void)fchmod(fd, 0644);                      // mkstemp makes it 0600; a cache is shared like any other output file
  std::string tmp = tmpl;
  FILE* f = fdopen(fd, "wb");
  if (!f) { close(fd); remove(tmp.c_str()); return pgpu_ctx_fail(ctx, PGPU_EINVAL, "cannot create the index file"); }
  const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(host.data(), 4, host.size(), f) == host.size();
  if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return pgpu_ctx_fail(ctx, PGPU_EDEVICE, "writing the index file failed"); }
  return PGPU_OK;
}

// PGPU_EINVAL when the file is missing, damaged or belongs to another sequence (the caller builds)
extern "C" int pgpu_index_load(pgpu_ctx* ctx, const char* path, const char* genomic, size_t len, pgpu_index** out) {
  if (!ctx || !path || !out || (len && !genomic)) return PGPU_EINVAL;
  *out = nullptr;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  FILE* f = fopen(path, "rb");
  if (!f) return PGPU_EINVAL;
  IndexFileHeader h;
  const size_t words = 2 * len + 1 + 2 * (size_t)KTAB_ENTRIES;
  std::vector<uint32_t> host;
  bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, INDEX_MAGIC, 8) == 0 && h.version == INDEX_VERSION && h.ktab == KTAB &&
            h.len == len && h.hash == fnv1a64(genomic, len);
  if (ok) { host.resize(words); ok = fread(host.data(), 4, words, f) == words && fgetc(f) == EOF; }
  fclose(f);
  if (ok) ok = h.payload_hash == words_hash(host.data(), words);
  // belt and braces: nothing that indexes the sequence or the suffix array may point outside
  for (size_t i = 0; ok && i < len; ++i) ok = host[i] < len;                                   // suffix array
  for (size_t i = 0; ok && i <= len; ++i) ok = host[len + i] <= len;                           // LCP
  for (size_t i = 0; ok && i < 2 * (size_t)KTAB_ENTRIES; ++i) ok = host[2 * len + 1 + i] <= len;   // k-mer intervals
  if (!ok) return PGPU_EINVAL;
  pgpu_index* idx = new (std::nothrow) pgpu_index();
  if (!idx) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of host memory");
  idx->len = len;
  hipStream_t st = pgpu_ctx_stream(ctx);
  const uint32_t n = (uint32_t)len;
  int rc = PGPU_OK;
  TRY_HIP(hipMalloc((void**)&idx->d_gen, len + 64));
  TRY_HIP(hipMemsetAsync(idx->d_gen, 0, len + 64, st));
  if (len) TRY_HIP(hipMemcpyAsync(idx->d_gen, genomic, len, hipMemcpyHostToDevice, st));
  TRY_HIP(upload_keys(idx, genomic, len, st));
  TRY_HIP(dmalloc(&idx->d_sa, n + 1)); TRY_HIP(dmalloc(&idx->d_lcp, n + 2));
  TRY_HIP(dmalloc(&idx->d_klo, KTAB_ENTRIES)); TRY_HIP(dmalloc(&idx->d_khi, KTAB_ENTRIES));
  if (n) TRY_HIP(hipMemcpyAsync(idx->d_sa, host.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  TRY_HIP(hipMemcpyAsync(idx->d_lcp, host.data() + n, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st));
  TRY_HIP(hipMemcpyAsync(idx->d_klo, host.data() + 2 * (size_t)n + 1, KTAB_ENTRIES * 4, hipMemcpyHostToDevice, st));
  TRY_HIP(hipMemcpyAsync(idx->d_khi, host.data() + 2 * (size_t)n + 1 + KTAB_ENTRIES, KTAB_ENTRIES * 4, hipMemcpyHostToDevice, st));
  TRY_HIP(build_lcf_tables(idx, genomic, st));
  TRY_HIP(hipStreamSynchronize(st));
done:
  if (rc != PGPU_OK) { hipFree(idx->d_gen); hipFree(idx->d_sa); hipFree(idx->d_lcp); hipFree(idx->d_klo); hipFree(idx->d_khi); hipFree(idx->d_key); hipFree(idx->d_focc); hipFree(idx->d_rmq); hipFree(idx->d_code); hipFree(idx->d_bad); delete idx; return rc; }
  *out = idx;
  return PGPU_OK;
}

extern "C" int pgpu_index_destroy(pgpu_ctx* ctx, pgpu_index* idx) {
  if (!ctx || !idx) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  hipStreamSynchronize(pgpu_ctx_stream(ctx));
  hipFree(idx->d_gen); hipFree(idx->d_sa); hipFree(idx->d_lcp); hipFree(idx->d_klo); hipFree(idx->d_khi); hipFree(idx->d_key);
  hipFree(idx->d_focc); hipFree(idx->d_rmq); hipFree(idx->d_code); hipFree(idx->d_bad);
  delete idx;
  return PGPU_OK;
}

extern "C" int pgpu_index_suffix_array(pgpu_ctx* ctx, const pgpu_index* idx, uint32_t* sa_out, size_t cap) {
  if (!ctx || !idx || !sa_out) return PGPU_EINVAL;
  if (cap < idx->len) return pgpu_ctx_fail(ctx, PGPU_ENOSPC, "suffix array buffer too small");
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  hipStream_t st = pgpu_ctx_stream(ctx);
  if (idx->len && (hipMemcpyAsync(sa_out, idx->d_sa, idx->len * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
                   hipStreamSynchronize(st) != hipSuccess))
    return pgpu_ctx_fail(ctx, PGPU_EDEVICE, "suffix array download failed");
  return PGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// C-ABI: pairings (plan = patterns resident in HBM; launch = all kernels; fetch = triples)
// ---------------------------------------------------------------------------------------------
struct pgpu_pairing_plan {
  const pgpu_index* idx = nullptr;
  size_t n_pat = 0;
  unsigned long long total_pos = 0;
  uint8_t* d_pats = nullptr;
  uint32_t *d_pcode = nullptr, *d_pbad = nullptr;     // the patterns at 2 bits per base + flag bits (PackedSeq)
  unsigned long long* d_pat_off = nullptr;
  uint32_t *d_lo = nullptr, *d_hi = nullptr, *d_a = nullptr, *d_thr = nullptr, *d_cnt = nullptr,
           *d_cnt_a = nullptr, *d_cnt_b = nullptr;
  unsigned long long *d_cand_off = nullptr, *d_out_off = nullptr, *d_out_first = nullptr;
  Cand* d_cand = nullptr;
  uint8_t* d_keep = nullptr;
  pgpu_pairing* d_out = nullptr;
  size_t cand_cap = 0, out_cap = 0;
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  unsigned long long n_cand = 0, n_out = 0;
  hipEvent_t ev[8] = {nullptr};
  float ms[7] = {0};
  bool pooled = false;
  // resident plan: only the patterns (bytes, offsets, packed) are the plan's own -- one allocation -- and every
  // buffer a run writes comes from the context's pool, shared with the other resident plans of the context
  bool resident = false;
  void* d_resident = nullptr;
  pgpu_ctx* owner = nullptr;
  // MEG stage (pgpu_meg.hip)
  void *d_meg_scratch = nullptr, *d_meg_info = nullptr;
  uint32_t* d_meg_bytes = nullptr;
  unsigned long long* d_meg_off = nullptr;
  uint8_t* d_meg_out = nullptr;
  size_t meg_cap = 0;
  unsigned long long meg_total = 0;
  hipEvent_t meg_ev[2] = {nullptr, nullptr};
  float meg_ms = 0.f;
  uint32_t last_L = 0;
  bool have_pairs = false;
};

// device buffer for a pairing plan: from the context's pool when the plan holds it
template <class T> static T* plan_alloc(pgpu_pairing_plan* p, int slot, size_t count) {
  const size_t bytes = (count ? count : 1) * sizeof(T);
  if (p->pooled) return (T*)pgpu_ctx_pool_get(p->owner, 1, slot, bytes);
  void* q = nullptr;
  if (hipMalloc(&q, bytes) != hipSuccess) return nullptr;
  pgpu_trace_alloc("plan device", q, bytes);
  return (T*)q;
}

static void pairing_plan_free(pgpu_pairing_plan* p) {
  if (!p) return;
  if (p->resident) { pgpu_ctx_pool_unshare(p->owner, 1, p); hipFree(p->d_resident); }
  else if (p->pooled) pgpu_ctx_pool_release(p->owner, 1);
  else {
    hipFree(p->d_pats); hipFree(p->d_pcode); hipFree(p->d_pbad); hipFree(p->d_pat_off); hipFree(p->d_lo); hipFree(p->d_hi); hipFree(p->d_a);
    hipFree(p->d_thr); hipFree(p->d_cnt); hipFree(p->d_cnt_a); hipFree(p->d_cnt_b);
    hipFree(p->d_cand_off); hipFree(p->d_out_off); hipFree(p->d_out_first); hipFree(p->d_cand);
    hipFree(p->d_keep); hipFree(p->d_out); hipFree(p->d_tmp);
    hipFree(p->d_meg_scratch); hipFree(p->d_meg_info); hipFree(p->d_meg_bytes); hipFree(p->d_meg_off); hipFree(p->d_meg_out);
  }
  for (auto& e : p->ev) if (e) hipEventDestroy(e);
  for (auto& e : p->meg_ev) if (e) hipEventDestroy(e);
  delete p;
}

// the buffers of a resident plan that a run writes: taken from the shared pool anew at every run (another
// plan may have made the pool grow, i.e. move, in between)
static bool resident_scratch(pgpu_pairing_plan* p) {
  const size_t tp = (size_t)p->total_pos, n_pat = p->n_pat;
  p->d_lo = plan_alloc<uint32_t>(p, 2, tp); p->d_hi = plan_alloc<uint32_t>(p, 3, tp);
  p->d_a = plan_alloc<uint32_t>(p, 4, tp); p->d_thr = plan_alloc<uint32_t>(p, 5, tp);
  p->d_cnt = plan_alloc<uint32_t>(p, 6, tp + 1); p->d_cnt_a = plan_alloc<uint32_t>(p, 7, tp + 1);
  p->d_cnt_b = plan_alloc<uint32_t>(p, 8, tp + 1);
  p->d_cand_off = plan_alloc<unsigned long long>(p, 9, tp + 1);
  p->d_out_off = plan_alloc<unsigned long long>(p, 10, tp + 1);
  p->d_out_first = plan_alloc<unsigned long long>(p, 11, n_pat + 1);
  p->d_tmp = plan_alloc<uint8_t>(p, 12, p->tmp_bytes);
  p->d_cand = nullptr; p->d_keep = nullptr; p->d_out = nullptr;              // sized (and taken) once the counts are known
  p->d_meg_scratch = nullptr; p->d_meg_out = nullptr;
  return p->d_lo && p->d_hi && p->d_a && p->d_thr && p->d_cnt && p->d_cnt_a && p->d_cnt_b && p->d_cand_off && p->d_out_off &&
         p->d_out_first && p->d_tmp;
}

static int pairing_plan_create(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns,
                               const uint64_t* pat_off, size_t n_pat, bool resident, pgpu_pairing_plan** out) {
  if (!ctx || !idx || !out || !pat_off || (n_pat && pat_off[n_pat] && !patterns)) return PGPU_EINVAL;
  *out = nullptr;
  for (size_t i = 0; i < n_pat; ++i)
    if (pat_off[i + 1] < pat_off[i] || pat_off[i + 1] - pat_off[i] > 0x7fffffffull) return pgpu_ctx_fail(ctx, PGPU_EINVAL, "bad pattern offsets");
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  pgpu_pairing_plan* p = new (std::nothrow) pgpu_pairing_plan();
  if (!p) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of host memory");
  int rc = PGPU_OK;
  hipStream_t st = pgpu_ctx_stream(ctx);
  p->idx = idx; p->n_pat = n_pat; p->total_pos = n_pat ? pat_off[n_pat] : 0;
  const size_t tp = (size_t)p->total_pos;
  p->owner = ctx;
  p->resident = resident && pgpu_ctx_pool_share(ctx, 1);
  p->pooled = p->resident || pgpu_ctx_pool_acquire(ctx, 1);
#define NEED(ptr) do { if (!(ptr)) { rc = pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of device memory (pairing plan)"); goto done; } } while (0)
  p->tmp_bytes = pgpu_scan_tmp_bytes(std::max(tp, n_pat) + 1);
  if (p->resident) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bad_words = (tp + 31) / 32;
    const size_t o_off = up(tp + 64), o_code = o_off + up((n_pat + 1) * sizeof(unsigned long long)),
                 o_bad = o_code + up((2 * bad_words + 4) * sizeof(uint32_t)), total = o_bad + up((bad_words + 4) * sizeof(uint32_t));
    TRY_HIP(hipMalloc(&p->d_resident, total));
    pgpu_trace_alloc("resident plan", p->d_resident, total);
    uint8_t* base = (uint8_t*)p->d_resident;
    p->d_pats = base; p->d_pat_off = (unsigned long long*)(base + o_off);
    p->d_pcode = (uint32_t*)(base + o_code); p->d_pbad = (uint32_t*)(base + o_bad);
    goto upload;
  }
  NEED(p->d_pats = plan_alloc<uint8_t>(p, 0, tp + 64));
  NEED(p->d_pat_off = plan_alloc<unsigned long long>(p, 1, n_pat + 1));
  NEED(p->d_lo = plan_alloc<uint32_t>(p, 2, tp)); NEED(p->d_hi = plan_alloc<uint32_t>(p, 3, tp));
  NEED(p->d_a = plan_alloc<uint32_t>(p, 4, tp)); NEED(p->d_thr = plan_alloc<uint32_t>(p, 5, tp));
  NEED(p->d_cnt = plan_alloc<uint32_t>(p, 6, tp + 1)); NEED(p->d_cnt_a = plan_alloc<uint32_t>(p, 7, tp + 1));
  NEED(p->d_cnt_b = plan_alloc<uint32_t>(p, 8, tp + 1));
  NEED(p->d_cand_off = plan_alloc<unsigned long long>(p, 9, tp + 1));
  NEED(p->d_out_off = plan_alloc<unsigned long long>(p, 10, tp + 1));
  NEED(p->d_out_first = plan_alloc<unsigned long long>(p, 11, n_pat + 1));
  NEED(p->d_tmp = plan_alloc<uint8_t>(p, 12, p->tmp_bytes));
  {
    const size_t bad_words = (tp + 31) / 32;
    NEED(p->d_pcode = plan_alloc<uint32_t>(p, 21, 2 * bad_words + 4));
    NEED(p->d_pbad = plan_alloc<uint32_t>(p, 22, bad_words + 4));
  }
upload:
  if (pgpu_ctx_timing(ctx)) for (auto& e : p->ev) TRY_HIP(hipEventCreate(&e));
  if (tp) TRY_HIP(hipMemcpyAsync(p->d_pats, patterns, tp, hipMemcpyHostToDevice, st));
  TRY_HIP(hipMemcpyAsync(p->d_pat_off, pat_off, (n_pat + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  TRY_HIP(pack_sequence(p->d_pats, tp, p->d_pcode, p->d_pbad, st));
  TRY_HIP(hipStreamSynchronize(st));
done:
  if (rc != PGPU_OK) { pairing_plan_free(p); return rc; }
  *out = p;
  return PGPU_OK;
}

extern "C" int pgpu_pairing_plan_create(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns,
                                        const uint64_t* pat_off, size_t n_pat, pgpu_pairing_plan** out) {
  return pairing_plan_create(ctx, idx, patterns, pat_off, n_pat, false, out);
}
extern "C" int pgpu_pairing_plan_create_resident(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns,
                                                 const uint64_t* pat_off, size_t n_pat, pgpu_pairing_plan** out) {
  return pairing_plan_create(ctx, idx, patterns, pat_off, n_pat, true, out);
}

// runs every kernel; the two size-dependent buffers (candidates, output) grow on demand, which
// costs one stream synchronisation after each scan (the totals are needed on the host anyway)
extern "C" int pgpu_pairing_plan_run(pgpu_ctx* ctx, pgpu_pairing_plan* p, const pgpu_pairing_params* params) {
  if (!ctx || !p || !params || params->min_factor_len == 0) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  int rc = PGPU_OK;
  hipStream_t st = pgpu_ctx_stream(ctx);
  const pgpu_index* ix = p->idx;
  const uint32_t n = (uint32_t)ix->len;
  const size_t tp = (size_t)p->total_pos;
  const PairParams prm{params->min_factor_len, params->min_string_depth_rate};
  const PackedSeq Tp{ix->d_code, ix->d_bad}, Pp{p->d_pcode, p->d_pbad};
  const dim3 pgrid((unsigned)(p->n_pat ? p->n_pat : 1)), pblk(64);
  p->n_cand = p->n_out = 0;
  p->have_pairs = false;
  if (p->n_pat == 0) return PGPU_OK;
  if (p->resident) {
    pgpu_ctx_pool_set_owner(ctx, 1, p);               // whatever another resident plan left there is gone from here on
    if (!resident_scratch(p)) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of device memory (pairing plan)");
  }
  pgpu_range_push("pairings");
  struct PopAtExit { ~PopAtExit() { pgpu_range_pop(); } } pop_at_exit;
  if (p->ev[0]) TRY_HIP(hipEventRecord(p->ev[0], st));
  hipLaunchKernelGGL(pair_locate_kernel, pgrid, pblk, 0, st, ix->d_gen, n, ix->d_sa, ix->d_klo, ix->d_khi, p->d_pats, p->d_pat_off, prm, p->d_lo, p->d_hi, p->d_a, Tp, Pp);
  if (p->ev[1]) TRY_HIP(hipEventRecord(p->ev[1], st));
  hipLaunchKernelGGL(pair_chain_kernel, dim3((unsigned)((p->n_pat + 63) / 64)), pblk, 0, st, ix->d_gen, n, ix->d_sa, ix->d_lcp,
                     p->d_pats, p->d_pat_off, (uint32_t)p->n_pat, prm, p->d_lo, p->d_hi, p->d_a, p->d_thr, Tp, Pp);
  if (p->ev[2]) TRY_HIP(hipEventRecord(p->ev[2], st));
  hipLaunchKernelGGL(pair_count_kernel, pgrid, pblk, 0, st, ix->d_gen, n, ix->d_sa, ix->d_key, ix->sigma, p->d_pats, p->d_pat_off, prm, p->d_lo, p->d_hi, p->d_thr, p->d_cnt, Tp, Pp);
  TRY_HIP(hipMemsetAsync(p->d_cnt + tp, 0, sizeof(uint32_t), st));
  pgpu_exclusive_scan_u32(p->d_cnt, p->d_cand_off, tp + 1, p->d_tmp, st);
  if (p->ev[3]) TRY_HIP(hipEventRecord(p->ev[3], st));
  TRY_HIP(hipMemcpyAsync(&p->n_cand, p->d_cand_off + tp, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  TRY_HIP(pgpu_ctx_wait(ctx));
  if (p->n_cand > p->cand_cap || !p->d_cand) {
    if (!p->pooled) { hipFree(p->d_cand); hipFree(p->d_keep); }
    p->d_cand = nullptr; p->d_keep = nullptr;
    p->cand_cap = (size_t)(p->n_cand + p->n_cand / 8 + 1024);
    NEED(p->d_cand = plan_alloc<Cand>(p, 13, p->cand_cap));
    NEED(p->d_keep = plan_alloc<uint8_t>(p, 14, p->cand_cap));
  }
  hipLaunchKernelGGL(pair_fill_kernel, pgrid, pblk, 0, st, ix->d_gen, n, ix->d_sa, ix->d_key, ix->sigma, p->d_pats, p->d_pat_off, prm, p->d_lo, p->d_hi, p->d_thr,
                     p->d_cand_off, p->d_cand, p->d_cnt_a, Tp, Pp);
  if (p->ev[4]) TRY_HIP(hipEventRecord(p->ev[4], st));
  hipLaunchKernelGGL(pair_cross_kernel, pgrid, pblk, 0, st, p->d_pat_off, p->d_cand_off, p->d_cand, p->d_cnt_a, p->d_keep, p->d_cnt_b);
  TRY_HIP(hipMemsetAsync(p->d_cnt_b + tp, 0, sizeof(uint32_t), st));
  pgpu_exclusive_scan_u32(p->d_cnt_b, p->d_out_off, tp + 1, p->d_tmp, st);
  if (p->ev[5]) TRY_HIP(hipEventRecord(p->ev[5], st));
  TRY_HIP(hipMemcpyAsync(&p->n_out, p->d_out_off + tp, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  TRY_HIP(pgpu_ctx_wait(ctx));
  if (p->n_out > p->out_cap || !p->d_out) {
    if (!p->pooled) hipFree(p->d_out);
    p->d_out = nullptr;
    p->out_cap = (size_t)(p->n_out + p->n_out / 8 + 1024);
    NEED(p->d_out = plan_alloc<pgpu_pairing>(p, 15, p->out_cap));
  }
  hipLaunchKernelGGL(pair_emit_kernel, pgrid, pblk, 0, st, p->d_pat_off, p->d_cand_off, p->d_cand, p->d_cnt_a, p->d_keep, p->d_out_off,
                     p->d_out, p->d_out_first, (uint32_t)p->n_pat, p->total_pos);
  if (p->ev[6]) TRY_HIP(hipEventRecord(p->ev[6], st));
  TRY_HIP(pgpu_ctx_wait(ctx));
  TRY_HIP(hipGetLastError());
  if (p->ev[0]) for (int k = 0; k < 6; ++k) hipEventElapsedTime(&p->ms[k], p->ev[k], p->ev[k + 1]);
  p->have_pairs = true; p->last_L = params->min_factor_len;
done:
  return rc;
}

// MEG of every pattern from the pairings that pgpu_pairing_plan_run left in HBM
extern "C" int pgpu_pairing_plan_run_meg(pgpu_ctx* ctx, pgpu_pairing_plan* p, const pgpu_meg_params* prm) {
  if (!ctx || !p || !prm || prm->min_factor_len == 0) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  p->meg_total = 0; p->meg_ms = 0.f;
  if (p->n_pat == 0) return PGPU_OK;
  if (!p->have_pairs || p->last_L != prm->min_factor_len)
    return pgpu_ctx_fail(ctx, PGPU_EINVAL, "run_meg needs the pairings of pgpu_pairing_plan_run with the same min_factor_len");
  if (p->resident && pgpu_ctx_pool_owner(ctx, 1) != p)
    return pgpu_ctx_fail(ctx, PGPU_EINVAL, "the pairings of this resident plan were overwritten by the run of another one");
  if (p->resident) { p->d_meg_scratch = nullptr; p->d_meg_out = nullptr; }
  int rc = PGPU_OK;
  hipStream_t st = pgpu_ctx_stream(ctx);
  const uint32_t np = (uint32_t)p->n_pat;
  pgpu_range_push("meg");
  struct PopAtExit { ~PopAtExit() { pgpu_range_pop(); } } pop_at_exit;
  if (!p->d_meg_scratch) {
    NEED(p->d_meg_scratch = plan_alloc<uint8_t>(p, 16, pgpu_meg_scratch_bytes(np)));
    NEED(p->d_meg_info = plan_alloc<uint8_t>(p, 17, (size_t)np * 16));
    NEED(p->d_meg_bytes = plan_alloc<uint32_t>(p, 18, (size_t)np + 1));
    NEED(p->d_meg_off = plan_alloc<unsigned long long>(p, 19, (size_t)np + 1));
    if (pgpu_ctx_timing(ctx) && !p->meg_ev[0]) for (auto& e : p->meg_ev) TRY_HIP(hipEventCreate(&e));
  }
  if (p->meg_ev[0]) TRY_HIP(hipEventRecord(p->meg_ev[0], st));
  pgpu_meg_launch_build(p->d_out, p->d_out_first, p->d_pat_off, np, prm, p->d_meg_scratch, p->d_meg_info, p->d_meg_bytes, st);
  TRY_HIP(hipMemsetAsync(p->d_meg_bytes + np, 0, sizeof(uint32_t), st));
  pgpu_exclusive_scan_u32(p->d_meg_bytes, p->d_meg_off, (size_t)np + 1, p->d_tmp, st);
  TRY_HIP(hipMemcpyAsync(&p->meg_total, p->d_meg_off + np, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  TRY_HIP(pgpu_ctx_wait(ctx));
  if (p->meg_total > p->meg_cap || !p->d_meg_out) {
    if (!p->pooled) hipFree(p->d_meg_out);
    p->d_meg_out = nullptr;
    p->meg_cap = (size_t)(p->meg_total + p->meg_total / 8 + 4096);
    NEED(p->d_meg_out = plan_alloc<uint8_t>(p, 20, p->meg_cap));
  }
  pgpu_meg_launch_emit(np, p->d_meg_scratch, p->d_meg_info, p->d_meg_off, p->d_meg_out, st);
  if (p->meg_ev[1]) TRY_HIP(hipEventRecord(p->meg_ev[1], st));
  TRY_HIP(pgpu_ctx_wait(ctx));
  TRY_HIP(hipGetLastError());
  if (p->meg_ev[0]) hipEventElapsedTime(&p->meg_ms, p->meg_ev[0], p->meg_ev[1]);
done:
  return rc;
}

extern "C" uint64_t pgpu_pairing_plan_meg_bytes(const pgpu_pairing_plan* p) { return p ? p->meg_total : 0; }
extern "C" double pgpu_pairing_plan_meg_ms(const pgpu_pairing_plan* p) { return p ? p->meg_ms : 0.0; }

extern "C" int pgpu_pairing_plan_fetch_meg(pgpu_ctx* ctx, pgpu_pairing_plan* p, void* out, size_t out_cap, uint64_t* rec_first) {
  if (!ctx || !p || !rec_first || (p->meg_total && !out)) return PGPU_EINVAL;
  if (out_cap < p->meg_total) return pgpu_ctx_fail(ctx, PGPU_ENOSPC, "MEG buffer too small");
  if (p->resident && p->n_pat && pgpu_ctx_pool_owner(ctx, 1) != p)
    return pgpu_ctx_fail(ctx, PGPU_EINVAL, "the results of this resident plan were overwritten by the run of another one");
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  int rc = PGPU_OK;
  hipStream_t st = pgpu_ctx_stream(ctx);
  if (p->n_pat == 0) { rec_first[0] = 0; return PGPU_OK; }
  if (p->meg_total) TRY_HIP(hipMemcpyAsync(out, p->d_meg_out, (size_t)p->meg_total, hipMemcpyDeviceToHost, st));
  TRY_HIP(hipMemcpyAsync(rec_first, p->d_meg_off, (p->n_pat + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  TRY_HIP(pgpu_ctx_wait(ctx));
done:
  return rc;
}

extern "C" int pgpu_host_alloc(pgpu_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return PGPU_EINVAL;
  *out = nullptr;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  if (hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of page-locked host memory");
  pgpu_trace_alloc("pgpu_host_alloc", *out, bytes);
  return PGPU_OK;
}
extern "C" int pgpu_host_free(pgpu_ctx* ctx, void* q) {
  if (!ctx) return PGPU_EINVAL;
  if (q && hipHostFree(q) != hipSuccess) return pgpu_ctx_fail(ctx, PGPU_EDEVICE, "hipHostFree failed");
  return PGPU_OK;
}

extern "C" uint64_t pgpu_pairing_plan_count(const pgpu_pairing_plan* p) { return p ? p->n_out : 0; }
extern "C" uint64_t pgpu_pairing_plan_positions(const pgpu_pairing_plan* p) { return p ? p->total_pos : 0; }
extern "C" double pgpu_pairing_plan_kernel_ms(const pgpu_pairing_plan* p, int k) { return (p && k >= 0 && k < 6) ? p->ms[k] : 0.0; }

extern "C" int pgpu_pairing_plan_fetch(pgpu_ctx* ctx, pgpu_pairing_plan* p, pgpu_pairing* out, size_t out_cap,
                                       uint64_t* out_first) {
  if (!ctx || !p || !out_first) return PGPU_EINVAL;
  if (out_cap < p->n_out) return pgpu_ctx_fail(ctx, PGPU_ENOSPC, "pairing buffer too small");
  if (p->resident && p->n_pat && pgpu_ctx_pool_owner(ctx, 1) != p)
    return pgpu_ctx_fail(ctx, PGPU_EINVAL, "the results of this resident plan were overwritten by the run of another one");
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  int rc = PGPU_OK;
  hipStream_t st = pgpu_ctx_stream(ctx);
  if (p->n_pat == 0) { out_first[0] = 0; return PGPU_OK; }
  if (p->n_out) TRY_HIP(hipMemcpyAsync(out, p->d_out, (size_t)p->n_out * sizeof(pgpu_pairing), hipMemcpyDeviceToHost, st));
  TRY_HIP(hipMemcpyAsync(out_first, p->d_out_first, (p->n_pat + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  TRY_HIP(pgpu_ctx_wait(ctx));
done:
  return rc;
}

extern "C" int pgpu_pairing_plan_destroy(pgpu_ctx* ctx, pgpu_pairing_plan* p) {
  if (!ctx || !p) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  hipStreamSynchronize(pgpu_ctx_stream(ctx));
  pairing_plan_free(p);
  return PGPU_OK;
}

extern "C" int pgpu_pairings(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns,
                             const uint64_t* pat_off, size_t n_pat, const pgpu_pairing_params* params,
                             pgpu_pairing* out, size_t out_cap, uint64_t* out_first, size_t* n_out) {
  pgpu_pairing_plan* p = nullptr;
  int rc = pgpu_pairing_plan_create(ctx, idx, patterns, pat_off, n_pat, &p);
  if (rc != PGPU_OK) return rc;
  rc = pgpu_pairing_plan_run(ctx, p, params);
  if (rc == PGPU_OK) {
    if (n_out) *n_out = (size_t)p->n_out;
    rc = pgpu_pairing_plan_fetch(ctx, p, out, out_cap, out_first);
  }
  pgpu_pairing_plan_destroy(ctx, p);
  return rc;
}
