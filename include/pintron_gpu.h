/*
 * pintron_gpu.h -- C-ABI of the MI355X (gfx950) est-fact accelerator library, libpintron_gpu.so.
 *
 * This is the drop-in boundary for PIntron's est-fact hot path: plain C, pointers and sizes only.
 * Each entry point names the reference routine(s) it replaces (paths relative to the AlgoLab/PIntron
 * tree).  The library has NO CPU fallback: every call fails with PGPU_EDEVICE when no gfx950
 * device is usable.
 *
 * Threading: a pgpu_ctx owns one HIP stream; calls on one context must be serialised by the
 * caller, different contexts are independent.  All functions return 0 (PGPU_OK) or a negative
 * errno-style code and never abort.
 */
#ifndef PINTRON_GPU_H
#define PINTRON_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGPU_OK       0
#define PGPU_EDEVICE (-5)    /* HIP runtime error / no device (EIO) */
#define PGPU_ENOMEM  (-12)
#define PGPU_EINVAL  (-22)
#define PGPU_ENOSPC  (-28)   /* caller's output buffer too small */
#define PGPU_ERANGE  (-34)   /* a job exceeds the supported dimensions (per-job status) */
#define PGPU_ENOSYS  (-38)   /* entry point not implemented in this build */

typedef struct pgpu_ctx pgpu_ctx;
typedef struct pgpu_index pgpu_index;
typedef struct pgpu_dp_plan pgpu_dp_plan;

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                    */
/* ------------------------------------------------------------------------------------------ */
int pgpu_init(int device, pgpu_ctx** ctx);
int pgpu_destroy(pgpu_ctx* ctx);
/* human-readable text of the last error on this context (never NULL) */
const char* pgpu_last_error(const pgpu_ctx* ctx);
/* ABI version of the loaded library */
int pgpu_abi_version(void);
/* HIP-event timing of every kernel group of the plans created afterwards (off by default; the
 * measurement hooks below return 0 without it) */
int pgpu_set_timing(pgpu_ctx* ctx, int enabled);
/* NUMA node of the host the context's GPU hangs off (its PCI device's numa_node), or -1 when the
 * platform does not say.  The est-fact host binds its threads to that node: on a two-socket host
 * that alone is worth 10-14 % (the per-EST logic is memory-latency bound and the batches cross
 * PCIe on that socket). */
int pgpu_device_numa_node(pgpu_ctx* ctx);
/* Profiler ranges (roctx): the library wraps the kernel groups of every plan ("dp_batch", "lcf",
 * "pairings", "meg", "index build") and the host program its phases ("est-fact step", "prefetch chunk k")
 * so that `rocprofv3 --marker-trace` shows them.  The reference has five wall-clock timers
 * (src/main-est-fact.c:95-99) and nothing per routine; SURVEY.md section 5 asks for both.  The roctx library is
 * loaded on first use and only when PGPU_MARKERS=1 or a rocprofiler tool is in the process; otherwise the
 * calls are two loads and a branch. */
void pgpu_range_push(const char* name);
void pgpu_range_pop(void);
/* identity of the loaded library: compiler, target and an FNV-1a hash of the code object bundle's build
 * stamp (__DATE__ __TIME__ + the compiler version) -- what smoke() prints so that a record can tell which
 * binary ran */
const char* pgpu_build_info(void);

/* ------------------------------------------------------------------------------------------ */
/* genomic index -- replaces lst_stree_new (stree_src/lst_stree.c:816) + preprocess_text /     */
/* stree_preprocess (src/aug_suffix_tree.c:68,247), called at src/main-est-fact.c:224-239.    */
/* The genomic sequence (after Ntails_removal) is copied to HBM once, with its suffix array.  */
/* ------------------------------------------------------------------------------------------ */
int pgpu_index_build(pgpu_ctx* ctx, const char* genomic, size_t len, pgpu_index** idx);
int pgpu_index_destroy(pgpu_ctx* ctx, pgpu_index* idx);
/* The index depends on the genomic sequence alone, and a gene is processed many times (re-runs of
 * the pipeline, parameter studies): save writes suffix array, LCP array and k-mer table to a file,
 * load brings them back into HBM without the construction.  load returns PGPU_EINVAL when the file
 * is missing, damaged or was made for another sequence (length + hash are checked) -- the caller
 * then builds.   */
int pgpu_index_save(pgpu_ctx* ctx, const pgpu_index* idx, const char* genomic, const char* path);
int pgpu_index_load(pgpu_ctx* ctx, const char* path, const char* genomic, size_t len, pgpu_index** idx);
/* copies the suffix array (len entries) back to the host; for tests and diagnostics */
int pgpu_index_suffix_array(pgpu_ctx* ctx, const pgpu_index* idx, uint32_t* sa_out, size_t cap);

/* ------------------------------------------------------------------------------------------ */
/* pairings -- replaces build_vertex_set (src/max-emb-graph.c:218-392): for every position p   */
/* of every pattern, the maximal pairings (p, t, l) of the pattern with the genomic, after the  */
/* two low-complexity filters, in the order of the reference's per-position lists.            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t min_factor_len;         /* config->min_factor_len (+ inc_pairing_len)  */
  uint32_t reserved;
  double   min_string_depth_rate;  /* config->min_string_depth_rate               */
} pgpu_pairing_params;

typedef struct { int32_t p, t, l; } pgpu_pairing;

/* patterns: concatenated pattern bytes; pat_off[i]..pat_off[i+1] delimits pattern i (n_pat+1
 * offsets).  out/out_cap: caller buffer for pairings; out_first[i]..out_first[i+1] delimits the
 * pairings of pattern i (n_pat+1 entries), sorted by (p, t, l) as pairing_compare orders them
 * (src/types.c:391-411).  Returns PGPU_ENOSPC (and the needed count in *n_out) when out_cap is
 * too small. */
int pgpu_pairings(pgpu_ctx* ctx, const pgpu_index* idx,
                  const char* patterns, const uint64_t* pat_off, size_t n_pat,
                  const pgpu_pairing_params* params,
                  pgpu_pairing* out, size_t out_cap, uint64_t* out_first, size_t* n_out);

/* The same in three steps for callers that keep the patterns resident (bench, batched host):
 * create uploads the patterns; run launches every kernel with the given parameters (may be
 * repeated with other parameters: the reference re-runs build_vertex_set with a longer
 * min_factor_len when a MEG is too complex, src/compute-est-fact.c:132-144); fetch downloads. */
typedef struct pgpu_pairing_plan pgpu_pairing_plan;
int pgpu_pairing_plan_create(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns,
                             const uint64_t* pat_off, size_t n_pat, pgpu_pairing_plan** plan);
/* The same plan (build_vertex_set, src/max-emb-graph.c:218-392, as above) for a batch that is cut into several
 * plans used one after the other (the chunks of the EST batch the host prefetches): only the patterns -- bytes, offsets, 2-bit packing -- are the plan's own, in ONE
 * device allocation; every buffer a run writes (16 of them, ~60 bytes per pattern position, and the MEG stage's
 * 11 KB per pattern) is shared by all resident plans of the context.  What a run leaves -- the pairings for
 * fetch / run_meg, the MEG records for fetch_meg -- is valid until another resident plan of the same context
 * runs; asking later is PGPU_EINVAL, never stale data.  (A one-shot process made 200 allocations of 7 GB for
 * its ten chunks before this: a third of its first step.) */
int pgpu_pairing_plan_create_resident(pgpu_ctx* ctx, const pgpu_index* idx, const char* patterns,
                                      const uint64_t* pat_off, size_t n_pat, pgpu_pairing_plan** plan);
int pgpu_pairing_plan_run(pgpu_ctx* ctx, pgpu_pairing_plan* plan, const pgpu_pairing_params* params);
uint64_t pgpu_pairing_plan_count(const pgpu_pairing_plan* plan);       /* pairings of the last run */
uint64_t pgpu_pairing_plan_positions(const pgpu_pairing_plan* plan);   /* pattern positions */
/* HIP-event time of stage k of the last run: 0 locate, 1 chain, 2 count+scan, 3 fill,
 * 4 cross+scan, 5 emit */
double pgpu_pairing_plan_kernel_ms(const pgpu_pairing_plan* plan, int k);
int pgpu_pairing_plan_fetch(pgpu_ctx* ctx, pgpu_pairing_plan* plan, pgpu_pairing* out, size_t out_cap,
                            uint64_t* out_first);
int pgpu_pairing_plan_destroy(pgpu_ctx* ctx, pgpu_pairing_plan* plan);

/* ------------------------------------------------------------------------------------------ */
/* maximal-embedding graphs -- for every pattern of a plan whose pairings have just been       */
/* computed (pgpu_pairing_plan_run), the rest of build_meg (src/compute-est-fact.c:101-131):    */
/* build_edge_set (src/max-emb-graph.c:650-676 with is_there_an_edge_strict :394-465,           */
/* add_edges_from :533-553, add_edges_from_source :555-599, add_edges_to_sink :601-647),        */
/* simplify_meg (src/meg-simplification.c:314,193-232,142-191), transitive_reduction            */
/* (:333-632), compact_short_edges (:258-312), is_too_complex_for_compaction and                */
/* is_too_complex (:68-139).  The pairings stay in HBM; what comes back is the finished graph.  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t min_factor_len;            /* the one the pairings were computed with            */
  int32_t  min_intron_length, max_intron_length;
  uint32_t max_pairings_in_MEG;
  double   max_prefix_discarded_rate, max_suffix_discarded_rate, max_freq_shortest_pairing;
  uint32_t trans_red, short_edge_comp;   /* booleans                                         */
} pgpu_meg_params;

#define PGPU_MEG_MAX_VERTICES 64      /* vertices ever created for one MEG (source and sink included) */
#define PGPU_MEG_MAX_DEGREE   32      /* out- or in-degree of a vertex                                 */
#define PGPU_MEG_TOO_COMPLEX  1u      /* too_complex of build_meg (:122-131): the caller retries        */
#define PGPU_MEG_UNAVAILABLE  2u      /* beyond the limits above (or cyclic): the record is a bare
                                         header and the caller builds this MEG from the pairings       */
/* One record per pattern, 4-byte aligned, little endian:
 *   u32 n_vertices, u32 n_edges, u32 flags, u32 0
 *   n_vertices x (i32 p, t, l)     in the order of the reference's position lists = the numbering
 *                                  meg_write prints (src/io-meg.c:161-170); [0] source, [last] sink
 *   (n_vertices + 1) x u16         first edge of each vertex (CSR)
 *   n_edges x u8                   target vertex, in the order of the reference's adjacency lists
 *   (pad to 4) u32 meg_text_len, u32 edges_text_len, then the two texts est-fact prints for this
 *                                  graph: meg_write's "(p,t,l)" lines, "#adj#", "i-j" lines
 *                                  (src/io-meg.c:146-190) and the lines of
 *                                  add_intronic_edges_to_file (src/max-emb-graph.c:677-699)
 * run_meg builds the records of all patterns (after pgpu_pairing_plan_run with the same
 * min_factor_len); meg_bytes = total size; fetch_meg copies them and the n_pat + 1 byte offsets. */
int pgpu_pairing_plan_run_meg(pgpu_ctx* ctx, pgpu_pairing_plan* plan, const pgpu_meg_params* params);
uint64_t pgpu_pairing_plan_meg_bytes(const pgpu_pairing_plan* plan);
int pgpu_pairing_plan_fetch_meg(pgpu_ctx* ctx, pgpu_pairing_plan* plan, void* out, size_t out_cap,
                                uint64_t* rec_first);
/* HIP-event time of the MEG kernels of the last run_meg (build + scan + emit) */
double pgpu_pairing_plan_meg_ms(const pgpu_pairing_plan* plan);

/* page-locked host memory for the buffers handed to the fetch calls (a pageable destination
 * makes the runtime stage the copy); plain malloc'ed memory works too, slower */
int pgpu_host_alloc(pgpu_ctx* ctx, size_t bytes, void** out);
int pgpu_host_free(pgpu_ctx* ctx, void* p);

/* ------------------------------------------------------------------------------------------ */
/* gather -- the single exchange of the EST-sharded run (one process per GPU):    */
/* each rank hands in the bytes it produced (packed factorization records, or the text of an   */
/* output file), rank 0 receives them in rank order = input order.  RCCL point-to-point over   */
/* xGMI; the library loads RCCL on first use.  The reference has no counterpart: its est-fact  */
/* is one process (src/main-est-fact.c:249-291 is the loop that is sharded here).              */
/* ------------------------------------------------------------------------------------------ */
typedef struct pgpu_comm pgpu_comm;
typedef struct { char bytes[128]; } pgpu_comm_id;     /* = ncclUniqueId */
/* rank 0 makes the id and passes it to the other ranks by any means (est-fact: a file) */
int pgpu_comm_unique_id(pgpu_ctx* ctx, pgpu_comm_id* id);
int pgpu_comm_init(pgpu_ctx* ctx, int rank, int world, const pgpu_comm_id* id, pgpu_comm** comm);
/* send/send_bytes: this rank's payload (host memory).  counts: `world` entries, filled on every
 * rank.  recv/recv_cap: rank 0 only, receives sum(counts) bytes.  Collective: all ranks call it. */
int pgpu_gather(pgpu_ctx* ctx, pgpu_comm* comm, const void* send, uint64_t send_bytes,
                void* recv, uint64_t recv_cap, uint64_t* counts);
/* fixed-size all-gather: every rank contributes `bytes` bytes (host memory) and receives world x bytes
 * in rank order.  est-fact --gpus=N uses it once per run for the ranks' status word and output sizes:
 * a rank that failed is seen by all before any payload moves. */
int pgpu_allgather(pgpu_ctx* ctx, pgpu_comm* comm, const void* send, uint64_t bytes, void* recv);
int pgpu_comm_destroy(pgpu_ctx* ctx, pgpu_comm* comm);

/* ------------------------------------------------------------------------------------------ */
/* batched dynamic programs                                                                   */
/* ------------------------------------------------------------------------------------------ */
enum pgpu_dp_kind {
  /* compute_alignment (src/compute-alignments.c:39-207): a = EST string, b = genomic string.
   * v[0]=score v[1]=alignment_dim; str[0], str[1] = offsets of EST_alignment and GEN_alignment in
   * the output string buffer (both NUL-terminated).
   * p0 = 1 / 2 (optional): the alignment is handle_endpoints' of a FIRST / LAST exon
   * (src/est-factorizations.c:2145,2204); p1, p2 = the complexity threshold's double bits.  The library may then
   * answer, beside the alignment, the exon check (KBAND with tail = 1) of the exon as that routine is going to trim
   * it (:2163-2196, :2231-2296): v[4] bit 3 set = answered, for the sub-operands v[2], v[3] (first exon: characters
   * of a / of b trimmed away at the front; last exon: characters of a / of b kept) with the bound v[4] >> 8;
   * v[4] bit 0 = K_band_edit_distance's verdict, bits 1-2 = the two dust comparisons.  The caller files the answer
   * under the question those values define and still does its own trimming: the extra answer saves a request when
   * the two agree, and is never used when they do not. */
  PGPU_DP_ALIGN = 0,
  /* compute_gap_alignment (src/refine-intron.c:560-890): a = EST window, b = genomic window.
   * v[0]=gap_alignment_dim v[1]=factor_cut v[2]=intron_start v[3]=intron_end
   * v[4]=intron_start_on_align v[5]=intron_end_on_align; strings at str[0], str[1] offsets. */
  PGPU_DP_GAP = 1,
  /* Levenshtein distance without N wildcard = last cell of edit_distance (src/refine.c:50-83)
   * = compute_edit_distance (src/compute-alignments.c:240-249).  v[0]=distance. */
  PGPU_DP_ED = 2,
  /* K_band_edit_distance (src/compute-alignments.c:319-453): p0 = upper_bound.
   * v[0]=returned bool, v[1]=*edit.
   * tail = 1 ("exon check", a = the exon on the genomic sequence, b = on the EST): also the two comparisons of
   * clean_low_complexity_exons_2 (src/est-factorizations.c:1687-1691) for this exon -- dustScore
   * (src/exon-complexity.c:50-78) of each operand against the threshold whose IEEE double bits are p1 (low word)
   * and p2 (high word): v[2] bit 0 = dust(a) > threshold, bit 1 = dust(b) > threshold.  The score is computed in
   * FP64 in the reference's operation order (integer sum, x 10.0, / (length - 2), / length). */
  PGPU_DP_KBAND = 3,
  /* find_longest_common_factor_dp (src/factorization-refinement.c:255-316): a = s1, b = s2.
   * v[0]=len v[1]=occ1 v[2]=occ2. */
  PGPU_DP_LCF = 4,
  /* general_refine_borders (src/refine.c:105-192): a = p, b = t, p0=min_p_cut p1=max_p_cut
   * p2=max_errs, tail = number (0..2) of valid bytes that follow t in its buffer (the reference
   * reads t[len_t], t[len_t+1] in getBursetFrequency_adaptor).
   * v[0]=returned bool v[1]=out_offset_p v[2]=out_offset_t1 v[3]=out_offset_t2 v[4]=edit. */
  PGPU_DP_BORDERS = 5,
  /* find_longest_affix (src/factorization-refinement.c:1136-1173): a = est, b = genomic.
   * v[0]=valid_cut v[1]=est cut v[2]=genomic cut. */
  PGPU_DP_AFFIX = 6,
  PGPU_DP_NKINDS = 7
};

#define PGPU_JOB_A_GENOMIC 1u   /* a_off indexes the resident genomic of the index, not the arena */
#define PGPU_JOB_B_GENOMIC 2u   /* same for b_off */

typedef struct {
  uint32_t kind;            /* enum pgpu_dp_kind */
  uint32_t flags;           /* PGPU_JOB_* */
  uint64_t a_off, b_off;    /* byte offsets of the operands (arena or genomic) */
  uint32_t a_len, b_len;
  uint32_t p0, p1, p2;      /* kind-specific parameters */
  uint32_t tail;            /* BORDERS only */
} pgpu_dp_job;              /* 48 bytes */

typedef struct {
  int32_t status;           /* PGPU_OK or PGPU_ERANGE for this job */
  int32_t v[6];             /* kind-specific, see enum */
  int32_t pad;
  uint64_t str[2];          /* ALIGN/GAP: offsets of the two alignment strings */
} pgpu_dp_result;           /* 48 bytes */

/* limits (per job); larger jobs get status PGPU_ERANGE */
#define PGPU_MAX_ROWS_LEV     65536u  /* ALIGN, AFFIX: a_len; ED, KBAND: min(a_len,b_len) (beyond 4096 rows
                                        one wave sweeps the matrix in strips of 4096 rows) */
#define PGPU_MAX_ROWS_BORDERS 4096u  /* BORDERS: a_len of the fast kernels; up to PGPU_MAX_ROWS_LEV a slow
                                        anti-diagonal kernel over HBM answers instead of refusing */
#define PGPU_MAX_ROWS_GAP     2048u  /* GAP: a_len of the fast kernels; larger windows take the slow kernel */
#define PGPU_MAX_GAP_SIDE    16000u  /* GAP: a_len and b_len */
#define PGPU_MAX_GAP_CELLS (1ull << 27)   /* GAP: (a_len+1)*(b_len+1), one direction byte per cell */
#define PGPU_MAX_COLS      1048576u

/* A plan holds a batch of jobs resident in HBM: operands, sorted job table, workspaces, results.
 * create = sort by kernel/size + one upload; launch = enqueue every kernel of the batch (fanned
 * over the context's streams, asynchronous) and, behind them, the download of results and
 * alignment strings; sync = wait; fetch = hand the results out in the caller's job order.
 * idx may be NULL when no job uses PGPU_JOB_*_GENOMIC.  One plan per context at a time is the
 * fast path (its device and pinned buffers are the context's own). */
int pgpu_dp_plan_create(pgpu_ctx* ctx, const pgpu_index* idx,
                        const pgpu_dp_job* jobs, size_t n_jobs,
                        const char* arena, size_t arena_len, pgpu_dp_plan** plan);
/* The same for a batch assembled from several producers (the host program's worker threads): the
 * jobs of a part address that part's arena; results come back in the order of the parts, jobs of
 * part 0 first.  Nothing is copied on the caller's side: the library gathers jobs and arenas
 * straight into its pinned upload image. */
typedef struct { const pgpu_dp_job* jobs; size_t n_jobs; const char* arena; size_t arena_len; } pgpu_dp_part;
int pgpu_dp_plan_create_parts(pgpu_ctx* ctx, const pgpu_index* idx, const pgpu_dp_part* parts, size_t n_parts,
                              pgpu_dp_plan** plan);
int pgpu_dp_plan_launch(pgpu_ctx* ctx, pgpu_dp_plan* plan);
int pgpu_dp_plan_sync(pgpu_ctx* ctx, pgpu_dp_plan* plan);
/* bytes the alignment strings of this plan need in fetch's `strings` buffer */
size_t pgpu_dp_plan_string_bytes(const pgpu_dp_plan* plan);
int pgpu_dp_plan_fetch(pgpu_ctx* ctx, pgpu_dp_plan* plan, pgpu_dp_result* results,
                       char* strings, size_t strings_cap);
/* copies the result table (n_jobs * sizeof(pgpu_dp_result), caller order) into DEVICE memory the
 * caller owns (e.g. a buffer handed to an RCCL gather); waits for the plan first (the table is
 * completed on the host: the LCF answers are decoded there from the keys the kernel leaves);
 * returns after the copy has completed */
int pgpu_dp_plan_results_to_device(pgpu_ctx* ctx, pgpu_dp_plan* plan, void* device_dst, size_t cap);
int pgpu_dp_plan_destroy(pgpu_ctx* ctx, pgpu_dp_plan* plan);

/* measurement hooks (bench.py): DP cells of the plan per kind with the reference's own loop
 * bounds, algorithmic HBM bytes per kind, and the duration of the last
 * launch of each kind's kernels measured with HIP events on the context's stream. */
uint64_t pgpu_dp_plan_cells(const pgpu_dp_plan* plan, int kind);
uint64_t pgpu_dp_plan_algo_bytes(const pgpu_dp_plan* plan, int kind);
double   pgpu_dp_plan_kernel_ms(const pgpu_dp_plan* plan, int kind);
uint64_t pgpu_dp_plan_launches(const pgpu_dp_plan* plan, int kind);

/* per-kernel-launch view of the same numbers: one group = one kernel launch of the plan */
typedef struct {
  char     name[48];        /* e.g. "lev_wave<ALIGN,R=4>", "align_traceback", "lcf" */
  int32_t  kind;            /* enum pgpu_dp_kind */
  int32_t  pad;
  uint64_t jobs, cells, algo_bytes;
  double   ms;              /* HIP-event duration of the last launch */
  double   t0_ms;           /* its start on the device's time line: milliseconds since an event the library recorded
                               when the first context of the process came up -- the same line for every context on
                               the device, so the launches of several contexts can be merged into the time the device
                               was busy (bench.py: kernel_busy_union_ms); -1 when timing is off */
} pgpu_group_info;
int pgpu_dp_plan_n_groups(const pgpu_dp_plan* plan);
int pgpu_dp_plan_group_info(const pgpu_dp_plan* plan, int i, pgpu_group_info* out);

/* one-shot convenience: create + launch + sync + fetch + destroy */
int pgpu_dp_batch(pgpu_ctx* ctx, const pgpu_index* idx,
                  const pgpu_dp_job* jobs, size_t n_jobs, const char* arena, size_t arena_len,
                  pgpu_dp_result* results, char* strings, size_t strings_cap,
                  size_t* strings_used);

#ifdef __cplusplus
}
#endif
#endif
