/* est-fact host program (C99): types and module interfaces.
 *
 * This is the host side of the MI355X est-fact: it mirrors the process contract and the
 * per-EST algorithm of PIntron's est-fact (src/main-est-fact.c, src/compute-est-fact.c) and
 * reaches the GPU only through include/pintron_gpu.h (pairings over the device index, batched
 * dynamic programs).  Citations are relative to the AlgoLab/PIntron tree.
 */
#ifndef ESTFACT_H
#define ESTFACT_H

#include <limits.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "ef_list.h"

/* ---- configuration (include/configuration.h:39-135, defaults src/options.ggo:94-370) -------- */
typedef struct {
  unsigned min_factor_len;
  int min_intron_length, max_intron_length;
  double min_string_depth_rate;
  double max_prefix_discarded_rate, max_suffix_discarded_rate;
  int max_prefix_discarded, max_suffix_discarded;
  unsigned max_site_difference;
  int max_number_of_factorizations;
  double max_coverage_diff;
  int max_exonNUM_diff, max_gapLength_diff;
  char retain_externals;
  unsigned max_pairings_in_MEG;
  double max_freq_shortest_pairing;
  int suffpref_length_on_est, suffpref_length_for_intron, suffpref_length_on_gen;
  bool trans_red, short_edge_comp;
  unsigned max_single_factorization_time;
  double complexity_threshold;
  char config_file[512];
} ef_config;

void ef_config_defaults(ef_config* c);
/* command line > config.ini > defaults (src/configuration.c:252-327); writes config-dump.ini.
 * Returns 0, or -1 after printing a message (invalid option / value out of range). */
int ef_config_load(ef_config* c, int argc, char** argv);

/* ---- sequences (include/types.h:140-196) ---------------------------------------------------- */
#define EF_POLYA_CHR '*'
#define EF_POLYT_CHR '#'

typedef struct {
  char* id;               /* FASTA header without '>' */
  char* seq;              /* working sequence: strand-corrected, polyA/T masked */
  char* original_seq;     /* output sequence (strand-corrected) */
  char* gb;               /* /gb= token or NULL */
  char* chr;              /* genomic only */
  char strand_as_read[16];
  int strand;
  bool fixed_strand;
  int abs_start, abs_end; /* genomic only */
  int pref_polyA_length, suff_polyA_length, pref_polyT_length, suff_polyT_length;
  int pref_N_length, suff_N_length;
} ef_seq;

/* read_multifasta (src/io-multifasta.c:133-164); returns number of records, -1 on I/O error */
long ef_read_multifasta(const char* path, ef_seq*** out);
void ef_seq_free(ef_seq* s);
void ef_parse_genomic_header(ef_seq* gen);                 /* :410-423 */
int  ef_ntails_removal(ef_seq* gen);                       /* :830-868; -1 when only N */
void ef_set_gb_identification(ef_seq* est);                /* :279-304 */
void ef_set_strand_and_rc(ef_seq* est);                    /* :425-504 */
void ef_reverse_and_complement(ef_seq* est);               /* :506-522 */
void ef_polyAT_substitution(ef_seq* est);                  /* :662-828 */
ef_seq* ef_copy_and_reverse(const ef_seq* est);            /* src/main-est-fact.c:67-87 */

/* ---- MEG (include/types.h:186-206) ---------------------------------------------------------- */
#define EF_SOURCE_START INT_MIN
#define EF_SINK_START   (INT_MAX - 200)
#define EF_SOURCE_LEN   200

typedef struct ef_pairing {
  int p, t, l;
  int id;
  bool visited;
  ef_list* adjs;
  ef_list* incs;
} ef_pairing;

typedef struct {
  size_t n;          /* |P| + 2: [0] = source, [1+i] = position i, [n-1] = sink */
  ef_list** v;
} ef_meg;

typedef struct { int32_t p, t, l; } ef_triple;

/* vertex set from the pairing triples of one pattern (already filtered and ordered as
 * build_vertex_set leaves them, src/max-emb-graph.c:218-392) */
ef_meg* ef_meg_from_pairings(const ef_triple* tr, size_t n_tr, size_t pattern_len);
void ef_meg_free(ef_meg* V);
void ef_build_edge_set(ef_meg* V, const ef_config* cfg);               /* src/max-emb-graph.c:650 */
void ef_simplify_meg(ef_meg* V, const ef_config* cfg);                 /* src/meg-simplification.c:314 */
void ef_transitive_reduction(ef_meg* V);                               /* :333,518 */
void ef_compact_short_edges(ef_meg* V, const ef_config* cfg);          /* :258 */
bool ef_is_too_complex_for_compaction(ef_meg* V);                      /* :68 */
bool ef_is_too_complex(ef_meg* V, const ef_config* cfg);               /* :89 */
void ef_meg_stats(ef_meg* V, size_t* pairings, size_t* edges);         /* :52 */
void ef_meg_write(FILE* f, ef_meg* V);                                 /* src/io-meg.c:146 */
void ef_intronic_edges_write(FILE* f, ef_meg* V);                      /* src/max-emb-graph.c:677 */

/* ---- backend: where pairings and dynamic programs are computed ------------------------------ */
typedef struct ef_backend {
  void* self;
  /* pairings of one pattern; *out is malloc'ed by the backend, freed by the caller */
  int (*pairings)(void* self, const char* pattern, size_t m, unsigned min_factor_len, double rate,
                  ef_triple** out, size_t* n);
} ef_backend;

/* build_meg (src/compute-est-fact.c:90-152): vertex set, edges, simplification, reduction,
 * compaction, complexity retry loop.  *inc_pairing_len is updated like the reference's. */
ef_meg* ef_build_meg(const ef_seq* est, ef_backend* be, const ef_config* shared_cfg, size_t* inc_pairing_len);

void ef_write_single_est_info(FILE* f, const ef_seq* s);              /* src/io-multifasta.c:270 */

#endif
