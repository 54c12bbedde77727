/*
 * oracle/dp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the dynamic-programming routines on PIntron's est-fact hot path.
 * It exists to CHECK the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg).  Nothing in pintron_amd/ or the C-ABI library links, loads or calls this code.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_vs_ref.py) against the
 * reference's own object code built from /root/reference by oracle/Makefile (oracle/_ref/), on
 * seeded random inputs and on DP calls captured from whole reference runs, and against the
 * reference's unit-test known answers for the Burset table (test/refine-intron_test.c:148-920)
 * committed as tests/golden/burset_known_answers.json.
 *
 * All citations are relative to /root/reference.
 */
#ifndef PINTRON_DP_ORACLE_H
#define PINTRON_DP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- global alignment with traceback: compute_alignment (src/compute-alignments.c:39-83) ---- */
/* est_aln / gen_aln must hold n+m+1 bytes.  Returns the score (M[n][m], 0 on the equal-string
 * shortcut :48-58); *dim = number of alignment columns. */
uint32_t orc_align(const char* est, size_t n, const char* gen, size_t m,
                   char* est_aln, char* gen_aln, int32_t* dim);
/* direction bytes exactly as ComputeAlignMatrix fills them (:85-147): dir[(i*m)+j], buffer
 * (n+1)*(m+1) zero-filled by the caller. Returns the score. */
uint32_t orc_align_matrix(const char* est, size_t n, const char* gen, size_t m, char* dir);

/* ---- Levenshtein distance without N wildcard ---- */
/* last cell of edit_distance (src/refine.c:50-83) == last cell of edit_distance_matrix
 * (src/compute-alignments.c:211-236).  No shortcut. */
uint32_t orc_edit_distance(const char* a, size_t la, const char* b, size_t lb);
/* full matrix in the layout of src/refine.c:50-83: rows = s2 (ls2+1), cols = s1 (ls1+1). */
void orc_edit_distance_full(const char* s1, size_t ls1, const char* s2, size_t ls2, uint32_t* M);
/* compute_best_suffix_cut / compute_best_prefix_cut (src/compute-alignments.c:252-313) */
uint32_t orc_best_suffix_cut(const char* s1, size_t l1, const char* s2, size_t l2,
                             uint32_t* cut1, uint32_t* cut2);
uint32_t orc_best_prefix_cut(const char* s1, size_t l1, const char* s2, size_t l2,
                             uint32_t* cut1, uint32_t* cut2);

/* ---- K-band edit distance: K_band_edit_distance (src/compute-alignments.c:319-453) ---- */
/* returns 1 when *edit <= upper_bound (the reference's bool), else 0 */
/* dustScore (src/exon-complexity.c:50-78) of s[0..len) */
double orc_dust_score(const char* s, size_t len);
/* the two comparisons of clean_low_complexity_exons_2 (src/est-factorizations.c:1687-1691) for one exon: bit 0 the
 * genomic side's score > threshold, bit 1 the EST side's */
uint32_t orc_dust_flags(const char* gen, size_t lg, const char* est, size_t le, double threshold);
int orc_kband(const char* seq1, size_t l1, const char* seq2, size_t l2, uint32_t upper_bound,
              uint32_t* edit);

/* ---- 3-state gap alignment: compute_gap_alignment (src/refine-intron.c:560-890) ---- */
typedef struct {
  int32_t dim;                    /* gap_alignment_dim */
  int32_t factor_cut;
  int32_t intron_start;
  int32_t intron_end;
  int32_t intron_start_on_align;
  int32_t intron_end_on_align;
  int32_t start_matrix;           /* 0 = L, 1 = G, 2 = R (:808-819) */
  int32_t score;                  /* value of the start matrix at (n,m) */
} orc_gap_result;
/* est_aln / gen_aln must hold n+m+1 bytes */
void orc_gap_align(const char* est, size_t n, const char* gen, size_t m,
                   char* est_aln, char* gen_aln, orc_gap_result* res);

/* ---- longest common factor with N wildcard (src/factorization-refinement.c:255-316) ---- */
void orc_lcf(const char* s1, size_t l1, const char* s2, size_t l2,
             uint32_t* occ1, uint32_t* occ2, uint32_t* len);

/* ---- Burset dinucleotide-pair frequency (src/refine-intron.c:362-556) ---- */
int orc_burset_frequency(const char* donor, const char* acceptor);       /* getBursetFrequency */
int orc_burset_adaptor(const char* t, size_t cut1, size_t cut2);          /* _adaptor :362-374 */

/* ---- general_refine_borders (src/refine.c:105-192) ---- */
typedef struct {
  uint32_t offset_p, offset_t1, offset_t2, edit_distance;
  int32_t ok;
} orc_borders_result;
void orc_refine_borders(const char* p, size_t len_p, size_t min_p_cut, size_t max_p_cut,
                        const char* t, size_t len_t, uint32_t max_errs,
                        orc_borders_result* res);

/* ---- find_longest_affix (src/factorization-refinement.c:1136-1173) ---- */
/* returns valid_cut; cuts untouched when 0 */
int orc_longest_affix(const char* est, size_t estl, const char* gen, size_t genl,
                      uint32_t* ecut, uint32_t* gcut);

/* number of DP cells the reference evaluates for a call (SURVEY.md section 8d accounting) */
uint64_t orc_cells_align(const char* est, size_t n, const char* gen, size_t m);
uint64_t orc_cells_kband(const char* s1, size_t l1, const char* s2, size_t l2, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
