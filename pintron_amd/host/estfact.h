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
#include <string.h>

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
  /* genomic only: positions of every upper-case ACGT 6-mer of seq, grouped by 6-mer code
   * (12 bits), ascending inside a group; built once by ef_seq_index_kmers */
  uint32_t* kmer_first;   /* 4097 offsets into kmer_pos */
  uint32_t* kmer_pos;
  /* genomic only: the positions of every byte of seq that is not an upper-case A, C, G or T, grouped by byte
   * value, ascending inside a group (a pattern that holds such a byte can only occur where the sequence has it) */
  uint32_t* other_first;  /* 257 offsets into other_pos */
  uint32_t* other_pos;
  /* genomic only: branch-point verdict of classify-intron per intron END position, filled on
   * demand (0 = not computed yet, 1 = no branch point, 2 = branch point found) */
  unsigned char* bps_memo;
  /* genomic only: MatInspector score of the four 5' splice-site matrices at every position
   * (classify-intron's GetScoreOf5Prime*BySS), filled once by ef_classify_prepare */
  double* score5_tab[4];
  size_t score5_len;
  /* genomic only: what classify-intron decides from the scores, per intron START (cls_start: the first two
   * characters' kind and, for the matrices that kind selects and for the general case, the two score
   * comparisons) and per intron END (cls_end: branch point found, last two characters' kind) -- two bytes
   * looked up per candidate intron instead of five doubles from four tables (ef_classify.c) */
  unsigned char* cls_start; unsigned char* cls_end;
  unsigned char in_arena;   /* the record and its strings lie in a record arena (ef_record_arena): ef_seq_free leaves them */
} ef_seq;

/* read_multifasta (src/io-multifasta.c:133-164); returns number of records, -1 on I/O error */
/* length of the genomic sequence: strlen() over 200 kb costs microseconds and the refinement code
 * asks for it several times per intron, so the last answer is kept per thread.  Only for the
 * genomic string, which is immutable for the whole run. */
/* Per-thread caches keyed on the genomic string's ADDRESS are only good while that genomic lives: a
 * process that runs several genes one after the other (est-fact --genes, direct mode) gets the next
 * gene's buffer at the same address from malloc.  Every load and every release of a genomic sequence
 * moves this epoch on, and a cache entry is valid for the epoch it was made in. */
extern unsigned ef_genomic_epoch;
static inline unsigned ef_genomic_epoch_now(void) { return __atomic_load_n(&ef_genomic_epoch, __ATOMIC_ACQUIRE); }
static inline void ef_genomic_epoch_bump(void) { __atomic_add_fetch(&ef_genomic_epoch, 1u, __ATOMIC_ACQ_REL); }
static inline size_t ef_genomic_len(const char* gen) {
  static _Thread_local const char* last; static _Thread_local size_t last_len; static _Thread_local unsigned last_epoch;
  const unsigned ep = ef_genomic_epoch_now();
  if (last != gen || last_epoch != ep) { last_len = strlen(gen); last = gen; last_epoch = ep; }
  return last_len;
}

/* Memory of the EST records of one input: a few large mappings (transparent huge pages where the system gives
 * them) filled by the parse and preparation threads, released as a whole.  Two million reads are eight million
 * small blocks and 600 000 page faults through malloc -- 1.4 of the 1.6 s a C5-sized file took to load. */
typedef struct ef_record_arena ef_record_arena;
ef_record_arena* ef_record_arena_new(void);
void ef_record_arena_free(ef_record_arena* a);
size_t ef_record_arena_regions(ef_record_arena* a, void** base, size_t* len, size_t max);
void ef_record_arena_enter(ef_record_arena* a);   /* the calling thread's records go to `a` from here on ... */
void ef_record_arena_leave(void);                 /* ... until here */

long ef_read_multifasta(const char* path, ef_seq*** out);
long ef_read_multifasta_part(const char* path, int part, int parts, ef_seq*** out);   /* the records of one rank of a sharded run */
long ef_read_multifasta_arena(const char* path, int part, int parts, ef_record_arena* arena, ef_seq*** out);
extern int ef_parse_threads;                  /* threads of the parse and of the preparation loop (8) */
void ef_seq_free(ef_seq* s);
void ef_seq_index_kmers(ef_seq* gen);                      /* used by the small-exon search */
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
  ef_list* emb_memo;      /* embeddings of the subtree rooted here, once computed */
} ef_pairing;

typedef struct ef_meg_s {
  size_t n;          /* |P| + 2: [0] = source, [1+i] = position i, [n-1] = sink */
  ef_list** v;
  /* the positions that ever held a vertex, ascending (a few dozen of the ~600): vertices are only
   * ever added at such a position, so visiting these in order = visiting all positions in order */
  size_t* act; size_t n_act;
  /* the device-built record this graph was made from (its two texts are printed as they are),
   * NULL for a graph built on the host */
  const void* rec;
  /* a graph made from a record is read-only and lives in ONE block (this structure first): position
   * table, list headers, vertices and list nodes are carved from it and freed with it */
  bool slab;
  /* ... and may be known by its record alone (v == NULL): statistics and the two texts come from the record, a
   * graph that is one path is enumerated from it too (ef_fact.c: chain_embedding); the lists are built when
   * somebody asks (ef_meg_lists) */
  struct ef_meg_s* lists;
} ef_meg;

/* first index k with act[k] >= lo */
static inline size_t ef_meg_act_from(const ef_meg* V, size_t lo) {
  size_t a = 0, b = V->n_act;
  while (a < b) { const size_t m = (a + b) / 2; if (V->act[m] < lo) a = m + 1; else b = m; }
  return a;
}
/* for (i = lo; i < hi; ++i) restricted to the positions that can hold vertices */
#define EF_MEG_FOR_POS(V, i, lo, hi) \
  for (size_t ef_k_ = ef_meg_act_from((V), (size_t)(lo)), i; \
       ef_k_ < (V)->n_act && (i = (V)->act[ef_k_]) < (size_t)(hi); ++ef_k_)

typedef struct { int32_t p, t, l; } ef_triple;

/* vertex set from the pairing triples of one pattern (already filtered and ordered as
 * build_vertex_set leaves them, src/max-emb-graph.c:218-392) */
ef_meg* ef_meg_from_pairings(const ef_triple* tr, size_t n_tr, size_t pattern_len);
/* the same structure from a device-built MEG record (vertices in position-list order + CSR) */
ef_meg* ef_meg_from_record(const void* rec, size_t pattern_len);
ef_meg* ef_meg_record_only(const void* rec, size_t pattern_len);      /* ... without the lists (see ef_meg.lists) */
ef_meg* ef_meg_lists(ef_meg* V);                                      /* V itself, or the graph with lists made for it */
extern int ef_endpoint_checks;     /* the end-exon alignments also ask for the trimmed exon's check (one suspension fewer per
                                      EST; PINTRON_ENDPOINT_CHECKS=0 switches it off.  Measured 3 % slower while the device
                                      routine was a call with a stack frame, even and + 4 % on a C5 share since it is inlined:
                                      DESIGN.md section 8) */
extern int ef_chain_fast_path;     /* PINTRON_CHAIN=0: every graph is enumerated through its lists (read once by ef_config_load) */
void ef_meg_free(ef_meg* V);
void ef_build_edge_set(ef_meg* V, const ef_config* cfg);               /* src/max-emb-graph.c:650 */
void ef_simplify_meg(ef_meg* V, const ef_config* cfg);                 /* src/meg-simplification.c:314 */
void ef_transitive_reduction(ef_meg* V);                               /* :333,518 */
void ef_compact_short_edges(ef_meg* V, const ef_config* cfg);          /* :258 */
bool ef_is_too_complex_for_compaction(ef_meg* V);                      /* :68 */
bool ef_is_too_complex(ef_meg* V, const ef_config* cfg);               /* :89 */
void ef_meg_stats(ef_meg* V, size_t* pairings, size_t* edges);         /* :52 */
struct ef_sink;
void ef_meg_write(struct ef_sink* f, ef_meg* V);                                 /* src/io-meg.c:146 */
void ef_intronic_edges_write(struct ef_sink* f, ef_meg* V);                      /* src/max-emb-graph.c:677 */

/* ---- output text ------------------------------------------------------------------------------
 * The record writers produce a few hundred short numeric fields per EST; they are assembled in a
 * small buffer with a hand-written integer formatter and handed to stdio in large pieces (printf
 * parsing was >10 % of the host time). */
/* where output text goes: a stream (sequential runs write the files as they go) or a growing
 * memory block (the batched runs keep the text of an EST until the files are written in order) */
typedef struct ef_sink { FILE* f; char* mem; size_t len, cap; } ef_sink;
static inline void ef_sink_write(ef_sink* s, const char* p, size_t n) {
  if (s->f) { fwrite(p, 1, n, s->f); return; }
  if (s->len + n > s->cap) { s->cap = (s->len + n) * 2 + 256; s->mem = (char*)realloc(s->mem, s->cap); }
  memcpy(s->mem + s->len, p, n); s->len += n;
}
static inline void ef_sink_puts(ef_sink* s, const char* str) { ef_sink_write(s, str, strlen(str)); }

/* A writer formats into [p, end): the tail of a memory sink itself (no staging copy), or a small
 * staging buffer in front of a stream.  efw_flush() publishes what has been written. */
typedef struct { ef_sink* s; char* p; char* end; char b[4096]; } ef_wbuf;
static inline void efw_room(ef_wbuf* w, size_t need);
static inline void efw_open(ef_wbuf* w, ef_sink* s) {
  w->s = s;
  if (s->f) { w->p = w->b; w->end = w->b + sizeof w->b; }
  else { w->p = w->end = NULL; efw_room(w, 128); }
}
static inline void efw_flush(ef_wbuf* w) {
  if (w->s->f) { if (w->p != w->b) { fwrite(w->b, 1, (size_t)(w->p - w->b), w->s->f); w->p = w->b; } }
  else if (w->p) w->s->len = (size_t)(w->p - w->s->mem);
}
/* at least `need` bytes at w->p (need <= 2048 for streams) */
static inline void efw_room(ef_wbuf* w, size_t need) {
  if (w->p && (size_t)(w->end - w->p) >= need) return;
  ef_sink* s = w->s;
  if (s->f) { efw_flush(w); return; }
  if (w->p) s->len = (size_t)(w->p - s->mem);
  if (s->len + need > s->cap) { s->cap = (s->len + need) * 2 + 256; s->mem = (char*)realloc(s->mem, s->cap); }
  w->p = s->mem + s->len; w->end = s->mem + s->cap;
}
static inline void efw_ch(ef_wbuf* w, char c) { efw_room(w, 1); *w->p++ = c; }
static inline void efw_mem(ef_wbuf* w, const char* s, size_t n) {
  if (w->s->f && n > 2048) { efw_flush(w); fwrite(s, 1, n, w->s->f); return; }
  efw_room(w, n);
  memcpy(w->p, s, n); w->p += n;
}
static inline void efw_str(ef_wbuf* w, const char* s) { efw_mem(w, s, strlen(s)); }
/* printf("%.*s"): at most `prec` characters, fewer when the string ends first */
static inline void efw_strn(ef_wbuf* w, const char* s, int prec) {
  if (prec <= 0) return;
  const char* z = (const char*)memchr(s, 0, (size_t)prec);          /* (vectorised: two exon strings per line of raw-multifasta-out) */
  efw_mem(w, s, z ? (size_t)(z - s) : (size_t)prec);
}
static inline void efw_int(ef_wbuf* w, long long v) {                     /* printf("%d") */
  char t[24]; int k = 0;
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
  efw_room(w, 24);
  if (v < 0) *w->p++ = '-';
  while (k) *w->p++ = t[--k];
}

/* ---- PINTRON_PROFILE=1: where the host time of the per-EST code goes -------------------------------
 * The per-EST code marks the phase it is in (one store and, when profiling is on, one time-stamp
 * read); the scheduler charges the cycles between two marks / suspensions of a fibre to the phase
 * that was current, and counts the suspensions and DP jobs asked in each phase.  Per thread; the
 * scheduler sums the threads of a step and prints the table (ef_sched.c). */
enum { EFP_OTHER = 0, EFP_MEG, EFP_EMBED, EFP_ENDPOINTS, EFP_EXTERNAL, EFP_DUST, EFP_NOISY, EFP_ADD, EFP_FILTERS,
       EFP_GAPERR, EFP_INTRON, EFP_TAIL, EFP_FACTREF, EFP_OUTPUT, EFP_SIDE, EFP_FREE, EFP_SCHED, EFP_SLEEP,
       EFP_REF_AFFIX, EFP_REF_FALSE_SMALL, EFP_REF_NEW_SMALL, EFP_REF_CLEAN, EFP_SCHED_LAUNCH, EFP_SCHED_COLLECT, EFP_SCHED_START, EFP_WAIT_PREFETCH,
       EFP_NS_PREFIX, EFP_NS_BETWEEN_ASK, EFP_NS_BETWEEN_CLASS, EFP_NS_BETWEEN_SEARCH, EFP_TMP1, EFP_TMP2, EFP_TMP3, EFP_N };
extern int ef_prof_on;
typedef struct { unsigned long long cyc[EFP_N], susp[EFP_N], jobs[EFP_N]; unsigned long long last; int dummy; int* cur;
                 unsigned long long ahead_hits, ahead_misses, ahead_asked, chain_graphs, other_graphs; } ef_prof_state;
extern _Thread_local ef_prof_state ef_prof;
static inline unsigned long long ef_prof_now(void) {
#if defined(__x86_64__)
  unsigned lo, hi; __asm__ volatile("rdtsc" : "=a"(lo), "=d"(hi)); return ((unsigned long long)hi << 32) | lo;
#else
  return 0;
#endif
}
/* enter phase `id`; returns the phase that was current (to go back to it) */
static inline int ef_phase(int id) {
  if (!ef_prof_on) return 0;
  ef_prof_state* p = &ef_prof;
  int* cur = p->cur ? p->cur : &p->dummy;
  const unsigned long long t = ef_prof_now();
  const int was = *cur;
  p->cyc[was] += t - p->last; p->last = t; *cur = id;
  return was;
}

/* ---- backend: where pairings and dynamic programs are computed ------------------------------ */
/* DP request/response in the vocabulary of include/pintron_gpu.h (same kinds, same result slots) */
enum { EF_DP_ALIGN = 0, EF_DP_GAP = 1, EF_DP_ED = 2, EF_DP_KBAND = 3, EF_DP_LCF = 4,
       EF_DP_BORDERS = 5, EF_DP_AFFIX = 6 };
typedef struct {
  int kind;
  const char* a; size_t la;      /* first operand  (EST side / s1 / p)       */
  const char* b; size_t lb;      /* second operand (genomic side / s2 / t)   */
  uint32_t p0, p1, p2, tail;     /* kind-specific, as in pgpu_dp_job         */
  uint32_t temp;                 /* an operand lies in a buffer of the caller's (not in the EST's or the genomic
                                    sequence): the question cannot be asked ahead (see ef_ahead) */
} ef_dp_req;                     /* operands that point into the genomic sequence are recognised
                                    by address and sent as PGPU_JOB_*_GENOMIC (no copy) */
typedef struct {
  int32_t v[6];
  char* s0; char* s1;            /* ALIGN/GAP: the two alignment rows, writable, in ONE block owned by
                                    the caller (ef_dp_res_release); each row is followed by at least
                                    EF_ROW_PAD zero bytes (the reference's scans run a little past
                                    the end of the rows) */
} ef_dp_res;
#define EF_ROW_PAD 64
/* room for two rows of up to `cap` characters each, zero-filled */
static inline void ef_dp_res_rows(ef_dp_res* r, size_t cap) {
  r->s0 = (char*)calloc(2 * (cap + EF_ROW_PAD), 1);
  r->s1 = r->s0 + cap + EF_ROW_PAD;
}
static inline void ef_dp_res_release(ef_dp_res* r) { free(r->s0); r->s0 = r->s1 = NULL; }

/* ---- questions asked ahead ---------------------------------------------------------------------
 * The per-EST code asks its dynamic programs where the reference calls them, one dependent step after the
 * other, and every question that goes to the device suspends the EST (ef_sched.c).  Many of them do not depend
 * on each other's answers -- the border refinements of all gaps of all candidate factorizations, the edit
 * distances and common factors of the small-exon searches of all introns -- so the code that is about to ask
 * them runs once in COLLECT mode first: it computes its questions from the current state exactly as it will
 * later, a question whose answer is not known yet is noted and the routine that asked returns without changing
 * anything; the noted questions then go out in ONE request, their answers stay with the EST, and the real run
 * finds them (ef_dp_many / ef_dp_one look here first).  A question is recognised by its identity -- kind,
 * operand addresses, lengths, parameters -- and only questions over the EST's and the genomic sequence
 * (immutable while the EST is processed; ef_dp_req.temp == 0) are kept, so an answer found here is the answer
 * the backend would give: a guess that turns out wrong (the state changed in between) costs a wasted job,
 * never a different result.  Answers with alignment rows (ALIGN, GAP) are not kept. */
#define EF_AHEAD_MAX 96
#define EF_DP_PENDING 1          /* collect mode: at least one answer is not known yet */
#define EF_AHEAD_SLOTS 256       /* hash slots of the kept answers (a power of two, > 2 x EF_AHEAD_MAX) */
typedef struct {
  int n, n_pending;
  bool collecting;
  unsigned char slot[EF_AHEAD_SLOTS];            /* entry + 1 by ef_req_hash (linear probing), 0 = free */
  ef_dp_req q[EF_AHEAD_MAX]; int32_t v[EF_AHEAD_MAX][6];     /* [0, n): kept answers; [n, n + n_pending): noted questions */
} ef_ahead;
static inline void ef_ahead_init(ef_ahead* ah) { ah->n = ah->n_pending = 0; ah->collecting = false; memset(ah->slot, 0, sizeof ah->slot); }

typedef struct ef_backend {
  void* self;
  /* pairings of one pattern; *out is malloc'ed by the backend, freed by the caller */
  int (*pairings)(void* self, const char* pattern, size_t m, unsigned min_factor_len, double rate,
                  ef_triple** out, size_t* n);
  /* one dynamic program; returns 0 or aborts the EST (never a CPU fallback in the product) */
  int (*dp)(void* self, const ef_dp_req* req, ef_dp_res* res);
  /* n dynamic programs that do not depend on each other, answered together (one suspension of the
   * EST instead of n); may be NULL, then ef_dp_many() asks one by one */
  int (*dp_many)(void* self, const ef_dp_req* reqs, ef_dp_res* res, size_t n);
  /* the finished MEG of `pattern` under `cfg`, when the backend has already built it on the
   * device behind the pairings (record layout: include/pintron_gpu.h, pgpu_pairing_plan_run_meg);
   * NULL (or a NULL hook) = not available, the caller builds the graph from the pairings */
  const void* (*meg)(void* self, const char* pattern, size_t m, const ef_config* cfg);
  ef_ahead* ahead;               /* the answers asked ahead for the EST being processed, or NULL */
} ef_backend;

static inline bool ef_req_same(const ef_dp_req* x, const ef_dp_req* y) {
  return x->kind == y->kind && x->a == y->a && x->b == y->b && x->la == y->la && x->lb == y->lb &&
         x->p0 == y->p0 && x->p1 == y->p1 && x->p2 == y->p2 && x->tail == y->tail;
}
static inline bool ef_req_keepable(const ef_dp_req* q) { return q->temp == 0 && q->kind != EF_DP_ALIGN && q->kind != EF_DP_GAP; }
static inline unsigned ef_req_hash(const ef_dp_req* q) {
  uint64_t h = (uint64_t)(uintptr_t)q->a * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uintptr_t)q->b * 0xC2B2AE3D27D4EB4Full;
  h ^= ((uint64_t)q->la << 32 | (uint64_t)q->lb) * 0x165667B19E3779F9ull + (uint64_t)q->kind;
  return (unsigned)(h >> 40) & (EF_AHEAD_SLOTS - 1);
}
static inline const int32_t* ef_ahead_find(const ef_ahead* ah, const ef_dp_req* q) {
  if (ah->n == 0 || !ef_req_keepable(q)) return NULL;
  for (unsigned s = ef_req_hash(q); ah->slot[s]; s = (s + 1) & (EF_AHEAD_SLOTS - 1))
    if (ef_req_same(&ah->q[ah->slot[s] - 1], q)) return ah->v[ah->slot[s] - 1];
  return NULL;
}
/* an answer that came with another request is kept too (only while nothing is noted: the noted questions sit behind
 * the kept ones) */
static inline unsigned ef_req_hash(const ef_dp_req* q);
static inline void ef_ahead_put(ef_ahead* ah, const ef_dp_req* q, const int32_t* v) {
  if (ah->n_pending || ah->n >= EF_AHEAD_MAX || !ef_req_keepable(q) || ef_ahead_find(ah, q)) return;
  ah->q[ah->n] = *q; memcpy(ah->v[ah->n], v, 6 * sizeof(int32_t));
  unsigned s = ef_req_hash(q);
  while (ah->slot[s]) s = (s + 1) & (EF_AHEAD_SLOTS - 1);
  ah->slot[s] = (unsigned char)(++ah->n);
}
/* the noted question at q[n] becomes a kept answer */
static inline void ef_ahead_keep_next(ef_ahead* ah, const int32_t* v) {
  memcpy(ah->v[ah->n], v, 6 * sizeof(int32_t));
  unsigned s = ef_req_hash(&ah->q[ah->n]);
  while (ah->slot[s]) s = (s + 1) & (EF_AHEAD_SLOTS - 1);
  ah->slot[s] = (unsigned char)(++ah->n);
}
static inline int ef_backend_ask(ef_backend* be, const ef_dp_req* reqs, ef_dp_res* res, size_t n) {
  if (be->dp_many) return be->dp_many(be->self, reqs, res, n);
  for (size_t k = 0; k < n; ++k) { const int rc = be->dp(be->self, &reqs[k], &res[k]); if (rc != 0) return rc; }
  return 0;
}
/* n independent questions: 0 = answered (res filled), EF_DP_PENDING (collect mode only) = noted, not answered,
 * anything else = the backend failed */
int ef_dp_many_ahead(ef_backend* be, const ef_dp_req* reqs, ef_dp_res* res, size_t n);     /* ef_fact.c */
static inline int ef_dp_many(ef_backend* be, const ef_dp_req* reqs, ef_dp_res* res, size_t n) {
  if (n == 0) return 0;
  memset(res, 0, n * sizeof(ef_dp_res));
  if (be->ahead && (be->ahead->n || be->ahead->collecting)) return ef_dp_many_ahead(be, reqs, res, n);
  return ef_backend_ask(be, reqs, res, n);
}
static inline int ef_dp_one(ef_backend* be, const ef_dp_req* req, ef_dp_res* res) { return ef_dp_many(be, req, res, 1); }
/* collect mode on / off; ef_ahead_flush asks what has been noted (one request) and keeps the answers */
static inline void ef_ahead_collect(ef_backend* be, bool on) { if (be->ahead) be->ahead->collecting = on; }
static inline bool ef_collecting(const ef_backend* be) { return be->ahead && be->ahead->collecting; }
int ef_ahead_flush(ef_backend* be);
extern int ef_ahead_on;            /* PINTRON_AHEAD=0 switches the asking ahead off (read once by ef_config_load) */

/* ---- factorizations (include/types.h:160-180) ------------------------------------------------ */
typedef struct { int EST_start, EST_end, GEN_start, GEN_end; } ef_factor;   /* 0-based inclusive */
/* a factorization = ef_list of ef_factor*; a list of factorizations = ef_list of ef_list* */

typedef struct {
  const ef_seq* info;
  ef_list* factorizations;       /* may be empty */
  ef_list* polyA_signals;        /* parallel lists of (void*)(intptr_t)0/1 */
  ef_list* polyadenil_signals;
} ef_est;

void ef_factorization_free(void* fact);
void ef_est_free(ef_est* e);

/* get_EST_factorizations (src/est-factorizations.c:126-594); NULL when the embedding enumeration
 * ran out of its work budget (the reference's timeout, made deterministic: see ef_fact.c) */
unsigned long long ef_work_high_water(void);
ef_est* ef_get_est_factorizations(const ef_seq* est, ef_meg* V, const ef_config* cfg,
                                  const ef_seq* gen, ef_backend* be);
/* refine_EST_factorizations & co (src/factorization-refinement.c) */
void ef_refine_est_factorizations(const ef_seq* gen, ef_est* e, const ef_config* cfg, ef_backend* be);
void ef_remove_factorizations_with_very_small_exons(ef_list* facts);
void ef_remove_duplicated_factorizations(ef_list* facts);
/* refine_intron (src/refine-intron.c:47-265) */
typedef struct {                 /* the two strings of one intron's gap alignment (:60-116) */
  char* seq_est; char* seq_gen; size_t le, lg;
  int dsl_est, dsl_gen, deleted_intron_dim;
  char buf_e[512], buf_g[512];
} ef_gap_window;
typedef struct { ef_gap_window w; ef_dp_res res; ef_factor donor, acceptor; } ef_gap_ahead;    /* ... asked before its turn (for these two exons), with the answer */
void ef_gap_window_build(const ef_config* cfg, const ef_seq* gen, const ef_seq* est, const ef_factor* donor,
                         const ef_factor* acceptor, ef_gap_window* w);
void ef_gap_window_release(ef_gap_window* w);
bool ef_refine_intron(const ef_config* cfg, const ef_seq* gen, const ef_seq* est, ef_factor* donor,
                      ef_factor* acceptor, bool first_intron, ef_backend* be, ef_gap_ahead* ahead);
/* intron classification (src/classify-intron.c:95): 0 = U12, 1 = U2, 2 = not classified */
int ef_classify_intron(const ef_seq* gen, int start, int end);
/* Burset frequencies (src/refine-intron.c:346-556) */
int ef_burset_frequency(const char* donor, const char* acceptor);
int ef_burset_adaptor(const char* t, size_t cut1, size_t cut2);
int ef_check_burset_patterns(const char* gen, int donor_left_on_gen, int acceptor_right_on_gen);
/* helpers shared by the modules */
char* ef_real_substring(int index, int length, const char* s);               /* src/util.c:138 */
ef_list* ef_clean_noisy_exons(ef_list* fact, const char* gen, const char* est, bool only_internals, ef_backend* be);
ef_list* ef_clean_external_exons(ef_list* fact, const char* gen, const char* est, ef_backend* be);
ef_list* ef_add_if_not_exists(ef_list* to_add, ef_list* list, const ef_config* cfg, bool* added);
uint32_t ef_edit_distance(ef_backend* be, const char* a, size_t la, const char* b, size_t lb);
/* compute_edit_distance (src/compute-alignments.c:240): equal strings short-cut on the host */
uint32_t ef_compute_edit_distance(ef_backend* be, const char* a, size_t la, const char* b, size_t lb);
/* write_multifasta_output (src/io-multifasta.c:187-246) */
void ef_write_multifasta_output(const ef_seq* gen, const ef_est* e, ef_sink* f, char retain_externals);
void ef_write_factorization_records(const ef_seq* gen, const ef_est* e, ef_sink* f, char retain_externals, uint32_t est_index);
/* ... and the text back from the records + the processed-ests text of the same ESTs (ef_records.c); 0, or -1 when they do not fit */
int ef_raw_text_from_records(const ef_seq* gen, const void* rec, size_t rl, const char* pests, size_t pl, ef_sink* out);
/* compute_est_fact (src/compute-est-fact.c:192-293) without the diagnostics side files */
typedef struct { ef_sink *fmeg, *fpmeg, *ftmeg, *fintronic; } ef_side_files;
ef_est* ef_compute_est_fact(const ef_seq* gen, const ef_seq* est, ef_backend* be, const ef_config* cfg,
                            const ef_side_files* side);

/* build_meg (src/compute-est-fact.c:90-152): vertex set, edges, simplification, reduction,
 * compaction, complexity retry loop.  *inc_pairing_len is updated like the reference's. */
ef_meg* ef_build_meg(const ef_seq* est, ef_backend* be, const ef_config* shared_cfg, size_t* inc_pairing_len);

void ef_write_single_est_info(ef_sink* f, const ef_seq* s);              /* src/io-multifasta.c:270 */

typedef struct { ef_config cfg; ef_seq* gen; ef_seq** list; size_t n; ef_record_arena* arena;
                 bool all_in_arena;   /* every record of `list` lies in `arena` (nothing to release one by one) */
                 unsigned char* has_rev;   /* per entry of `list` (or NULL): the next entry is its reverse-complement sibling */
} ef_inputs;
typedef struct { FILE* flog; ef_sink fout, fests, fmeg, fpmeg, ftmeg, fintronic; ef_side_files side; } ef_outputs;
int ef_load_inputs(int argc, char** argv, ef_inputs* in);     /* = ef_load_genomic + ef_load_ests */
int ef_load_genomic(int argc, char** argv, ef_inputs* in);    /* = ef_load_genomic_sequence + ef_prepare_genomic_tables */
int ef_load_genomic_sequence(int argc, char** argv, ef_inputs* in);
void ef_prepare_genomic_tables(ef_inputs* in);
int ef_load_ests(ef_inputs* in);
extern int ef_shard_rank, ef_shard_world;   /* ef_load_ests keeps the rank's range of the input ESTs */
void ef_free_inputs(ef_inputs* in);
int ef_open_outputs(ef_outputs* o);
void ef_info_mark(const char* label);       /* a phase boundary for info-pid-<pid>.log (ef_estfact.c) */
void ef_close_outputs(ef_outputs* o);
void ef_classify_init(void);     /* loads the PWM tables once (call before threads start) */
/* per-gene tables of the intron classifier: the branch-point verdict of every intron end and the
 * 5' splice-site scores of every intron start depend on the genomic sequence alone, so they are
 * computed once per gene (a few threads, milliseconds) instead of once per candidate intron */
void ef_classify_prepare(ef_seq* gen);

/* the reference's closing stderr lines: five timers, "End", resource usage (ef_estfact.c) */
void ef_log_reference_timers(double suffix_tree_s, double algorithm_s, double compositions_s, double io_s, double total_s);
/* the whole est-fact process (src/main-est-fact.c:90-339); the caller supplies the backend */
int ef_run(int argc, char** argv, ef_backend* (*open_backend)(const ef_seq* gen), void (*close_backend)(ef_backend*));

#endif
