/* compute_est_fact and the est-fact process flow.
 * Behaviour follows src/compute-est-fact.c:192-293 and src/main-est-fact.c:90-339 of the reference
 * (same files in cwd, same record formats, same order of ESTs and of the reverse-complement
 * siblings).  The wall-clock timeout of the reference (src/my_time.c:177-198) is a deterministic
 * work budget here (ef_fact.c); its retry branch is the reference's. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>

#include "estfact.h"

/* write_multifasta_output (src/io-multifasta.c:187-246) */
void ef_write_multifasta_output(const ef_seq* gen, const ef_est* e, ef_sink* f, char retain_externals) {
  if (!e->factorizations || efl_empty(e->factorizations)) return;
  ef_iter fi = efl_begin(e->factorizations), pa = efl_begin(e->polyA_signals), pd = efl_begin(e->polyadenil_signals);
  while (efi_has_next(&fi)) {
    ef_list* fact = (ef_list*)efi_next(&fi);
    int polya = efi_next(&pa) == (void*)1, polyad = efi_next(&pd) == (void*)1;
    const size_t n = efl_size(fact);
    if (!(retain_externals || (n > 2 || (n == 2 && e->info->suff_polyA_length != -1)))) continue;
    ef_wbuf w; efw_open(&w, f);
    efw_ch(&w, '>'); efw_str(&w, e->info->id); efw_ch(&w, '\n');
    if (!retain_externals) { polya = 0; polyad = 0; }
    efw_str(&w, "#polya="); efw_int(&w, polya); efw_str(&w, "\n#polyad="); efw_int(&w, polyad); efw_ch(&w, '\n');
    unsigned counter = 1;
    const unsigned l_index = retain_externals == 0 ? 1 : 0;
    const unsigned r_index = retain_externals == 0 ? (e->info->suff_polyA_length == -1 ? (unsigned)n : (unsigned)n + 1) : (unsigned)n + 1;
    ef_iter xi = efl_begin(fact);
    while (efi_has_next(&xi)) {
      const ef_factor* x = (const ef_factor*)efi_next(&xi);
      if (counter > l_index && counter < r_index) {        /* "%d %d %d %d %.*s %.*s\n" */
        efw_int(&w, x->EST_start + 1); efw_ch(&w, ' '); efw_int(&w, x->EST_end + 1); efw_ch(&w, ' ');
        efw_int(&w, gen->pref_N_length + x->GEN_start + 1); efw_ch(&w, ' ');
        efw_int(&w, gen->pref_N_length + x->GEN_end + 1); efw_ch(&w, ' ');
        efw_strn(&w, e->info->original_seq + x->EST_start, x->EST_end + 1 - x->EST_start); efw_ch(&w, ' ');
        efw_strn(&w, gen->original_seq + gen->pref_N_length + x->GEN_start, x->GEN_end + 1 - x->GEN_start);
        efw_ch(&w, '\n');
      }
      ++counter;
    }
    efw_flush(&w);
  }
}

/* The same factorizations as packed binary records, for consumers that do not want to parse text
 * (min-factorization reads "%d %d %d %d", #polya=, #polyad= and groups by
 * header, src/io-factorizations.c:128-231).  Little-endian, per aligned EST (= per entry of processed-ests.txt):
 *   u32 est_index (position in ests.txt), u32 n_factorizations, then per factorization
 *   u8 polya, u8 polyad, u16 n_exons, then per exon 4 x i32: EST_start, EST_end, GEN_start, GEN_end
 *   exactly as printed (1-based, genomic coordinates shifted by the removed N prefix).
 * The selection rules (retain_externals, polyA suffix) are those of the text writer above. */
void ef_write_factorization_records(const ef_seq* gen, const ef_est* e, ef_sink* f, char retain_externals, uint32_t est_index) {
  if (!e->factorizations || efl_empty(e->factorizations)) return;
  uint32_t n_fact = 0;
  size_t cap = 256, len = 8;
  unsigned char* b = (unsigned char*)malloc(cap);
  ef_iter fi = efl_begin(e->factorizations), pa = efl_begin(e->polyA_signals), pd = efl_begin(e->polyadenil_signals);
  while (efi_has_next(&fi)) {
    ef_list* fact = (ef_list*)efi_next(&fi);
    int polya = efi_next(&pa) == (void*)1, polyad = efi_next(&pd) == (void*)1;
    const size_t n = efl_size(fact);
    if (!(retain_externals || (n > 2 || (n == 2 && e->info->suff_polyA_length != -1)))) continue;
    if (!retain_externals) { polya = 0; polyad = 0; }
    const unsigned l_index = retain_externals == 0 ? 1 : 0;
    const unsigned r_index = retain_externals == 0 ? (e->info->suff_polyA_length == -1 ? (unsigned)n : (unsigned)n + 1) : (unsigned)n + 1;
    if (len + 4 + 16 * n > cap) { cap = (len + 4 + 16 * n) * 2; b = (unsigned char*)realloc(b, cap); }
    unsigned char* head = b + len;
    len += 4;
    uint16_t n_exons = 0;
    unsigned counter = 1;
    ef_iter xi = efl_begin(fact);
    while (efi_has_next(&xi)) {
      const ef_factor* x = (const ef_factor*)efi_next(&xi);
      if (counter > l_index && counter < r_index) {
        const int32_t v[4] = { x->EST_start + 1, x->EST_end + 1, gen->pref_N_length + x->GEN_start + 1, gen->pref_N_length + x->GEN_end + 1 };
        memcpy(b + len, v, 16); len += 16;
        ++n_exons;
      }
      ++counter;
    }
    head[0] = (unsigned char)polya; head[1] = (unsigned char)polyad; memcpy(head + 2, &n_exons, 2);
    ++n_fact;
  }
  /* one group per ALIGNED EST = per entry of processed-ests.txt, also when --retain-externals=false left none of
   * its factorizations to print (n_factorizations 0): the groups and the entries of that file stay in step, which
   * is what lets a reader print the text from the two (ef_records.c) */
  memcpy(b, &est_index, 4); memcpy(b + 4, &n_fact, 4);
  ef_sink_write(f, (const char*)b, len);
  free(b);
}

static unsigned long long mono_us(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (unsigned long long)ts.tv_sec * 1000000ull + (unsigned long long)(ts.tv_nsec / 1000);
}

/* compute_est_fact (src/compute-est-fact.c:192-293) */
ef_est* ef_compute_est_fact(const ef_seq* gen, const ef_seq* est, ef_backend* be, const ef_config* cfg,
                            const ef_side_files* side) {
  size_t inc = 0, prev_tp = 0, prev_te = 0, tp, te;
  ef_est* fe = NULL;
  bool expired;
  ef_ahead ahead;                                /* answers asked ahead for this EST (estfact.h) */
  if (ef_ahead_on) ef_ahead_init(&ahead);
  be->ahead = ef_ahead_on ? &ahead : NULL;
  do {
    ef_meg* V = NULL;
    bool same;
    const unsigned long long t_meg0 = mono_us();
    do {
      ef_phase(EFP_MEG);
      V = ef_build_meg(est, be, cfg, &inc);
      ef_meg_stats(V, &tp, &te);
      same = prev_tp > 2 && prev_te > 0 && (prev_tp <= tp || prev_te <= te);
      if (same) { ++inc; ef_meg_free(V); }
    } while (same);
    prev_tp = tp; prev_te = te;
    const unsigned long long t_meg1 = mono_us();
    /* internal_get_EST_factorizations (:154-190); NULL = budget spent ("timeout expired") */
    fe = ef_get_est_factorizations(est, V, cfg, gen, be);
    expired = fe == NULL;
    if (fe) {
      ef_phase(EFP_FACTREF);
      ef_refine_est_factorizations(gen, fe, cfg, be);
      ef_remove_factorizations_with_very_small_exons(fe->factorizations);
      if (!efl_empty(fe->factorizations)) ef_remove_duplicated_factorizations(fe->factorizations);
    }
    const bool aligned = fe && !efl_empty(fe->factorizations);
    ef_phase(EFP_SIDE);
    size_t meg_text_at = 0, meg_text_len = 0;                    /* the record + graph just printed, when it lies in memory */
    if ((!expired || aligned) && side && side->fmeg) {            /* report_meg (:73-88) */
      ef_sink_puts(side->fmeg, "\n\n***********\n\n");
      meg_text_at = side->fmeg->len;
      ef_write_single_est_info(side->fmeg, est);
      ef_meg_write(side->fmeg, V);
      if (side->fmeg->f) fflush(side->fmeg->f);
      else meg_text_len = side->fmeg->len - meg_text_at;
    }
    if (aligned && side) {
      if (side->fintronic) {
        ef_sink_puts(side->fintronic, ">"); ef_sink_puts(side->fintronic, est->id); ef_sink_puts(side->fintronic, "\n");
        ef_intronic_edges_write(side->fintronic, V);
      }
      /* processed-megs.txt repeats, for an aligned EST, what megs.txt has just got: the same two writers over the same
       * record and graph -- copied instead of formatted again */
      if (side->fpmeg) {
        if (meg_text_len && !side->fpmeg->f) ef_sink_write(side->fpmeg, side->fmeg->mem + meg_text_at, meg_text_len);
        else { ef_write_single_est_info(side->fpmeg, est); ef_meg_write(side->fpmeg, V); }
      }
      /* "<meg us> <composition us> <#factorizations>" (src/compute-est-fact.c:265-268: the intervals of its
       * two per-EST timers): the microseconds this EST spent from asking for its MEG to having it, and from there
       * to its refined factorizations -- in the batched program the second includes the time the EST waited for
       * the device's answers among the other ESTs */
      if (side->ftmeg) {
        ef_wbuf w; efw_open(&w, side->ftmeg);         /* "%llu %llu %zu\n" */
        efw_int(&w, (long long)(t_meg1 - t_meg0)); efw_ch(&w, ' '); efw_int(&w, (long long)(mono_us() - t_meg1)); efw_ch(&w, ' ');
        efw_int(&w, (long long)efl_size(fe->factorizations)); efw_ch(&w, '\n');
        efw_flush(&w);
      }
    }
    if (expired) ++inc;                                          /* :277-283: longer factors, again */
    ef_phase(EFP_FREE);
    ef_meg_free(V);
    ef_phase(EFP_OTHER);
  } while (expired);
  be->ahead = NULL;
  return fe;
}

typedef struct { ef_seq** ests; ef_seq** revs; long lo, hi; ef_record_arena* arena; bool all_in_arena; } prep_job;
static void* prep_main(void* arg) {
  prep_job* j = (prep_job*)arg;
  if (j->arena) ef_record_arena_enter(j->arena);          /* the reversed siblings and the /gb= names */
  bool all = true;
  for (long i = j->lo; i < j->hi; ++i) {
    ef_seq* est = j->ests[i];
    ef_set_gb_identification(est);
    ef_set_strand_and_rc(est);
    ef_polyAT_substitution(est);
    all = all && est->in_arena;
    if (!est->fixed_strand) {
      ef_seq* rev = ef_copy_and_reverse(est);
      ef_polyAT_substitution(rev);
      j->revs[i] = rev;
      all = all && rev->in_arena;
    }
  }
  j->all_in_arena = all;
  if (j->arena) ef_record_arena_leave();
  return NULL;
}

/* inputs of one est-fact run: configuration, genomic, prepared EST list (siblings interleaved) */
int ef_load_inputs(int argc, char** argv, ef_inputs* in) {
  const int rc = ef_load_genomic(argc, argv, in);
  return rc ? rc : ef_load_ests(in);
}

/* first half: configuration + genomic.txt (enough to start building the index) */
int ef_load_genomic(int argc, char** argv, ef_inputs* in) {
  const int rc = ef_load_genomic_sequence(argc, argv, in);
  if (rc == 0) ef_prepare_genomic_tables(in);
  return rc;
}

/* ... in two steps for the batched program: the sequence as the device index needs it (read, header, N tails:
 * milliseconds), and the host-side tables over it (k-mer positions of the small-exon search, the four score
 * tables of the intron classifier: 0.3 s for a 1 Mb gene on eight threads) -- the second runs beside the
 * start-up of the GPU runtime instead of in front of it */
void ef_prepare_genomic_tables(ef_inputs* in) {
  ef_seq_index_kmers(in->gen);
  ef_classify_prepare(in->gen);
}

int ef_load_genomic_sequence(int argc, char** argv, ef_inputs* in) {
  memset(in, 0, sizeof(*in));
  if (ef_config_load(&in->cfg, argc, argv) != 0) return 2;
  ef_seq** gens = NULL;
  const long ng = ef_read_multifasta("genomic.txt", &gens);
  if (ng < 0) { fprintf(stderr, "* FATAL File genomic.txt not found! Terminating\n"); return 1; }
  if (ng != 1) { fprintf(stderr, "* FATAL genomic.txt must hold exactly one sequence\n"); return 1; }
  in->gen = gens[0];
  free(gens);
  ef_parse_genomic_header(in->gen);
  if (ef_ntails_removal(in->gen) != 0) { fprintf(stderr, "* FATAL The sequence is only composed by Ns.\n"); return 1; }
  return 0;
}

/* EST-sharded runs (one process per GPU): rank r of w keeps a contiguous range of the input ESTs,
 * balanced by sequence length; an EST and its reverse-complement sibling are one input record,
 * so they stay together (src/main-est-fact.c:266-284 tries the sibling only when the EST fails) */
int ef_shard_rank = 0, ef_shard_world = 1;

/* second half: ests.txt and the preparation of every EST */
int ef_load_ests(ef_inputs* in) {
  ef_seq** ests = NULL;
  /* EST-sharded run: the rank's contiguous part of the file, cut by bytes at record starts (the sequences are
   * what the bytes are, so the parts are balanced by total sequence length up to the headers) */
  in->arena = ef_record_arena_new();
  long n_in = ef_read_multifasta_arena("ests.txt", ef_shard_rank, ef_shard_world, in->arena, &ests);
  if (n_in < 0) { fprintf(stderr, "* FATAL File ests.txt not found! Terminating\n"); return 1; }
  /* preparation loop (src/main-est-fact.c:190-213): every EST is prepared on its own, so the
   * loop is split over a few threads; the list is then filled in input order */
  ef_seq** revs = (ef_seq**)calloc((size_t)n_in + 1, sizeof(ef_seq*));
  enum { PREP_THREADS_MAX = 32 };
  const int prep_threads = ef_parse_threads < 1 ? 1 : ef_parse_threads > PREP_THREADS_MAX ? PREP_THREADS_MAX : ef_parse_threads;
  prep_job jobs[PREP_THREADS_MAX]; pthread_t th[PREP_THREADS_MAX]; bool started[PREP_THREADS_MAX];
  for (int t = 0; t < prep_threads; ++t) {
    jobs[t].ests = ests; jobs[t].revs = revs; jobs[t].arena = in->arena;
    jobs[t].lo = n_in * t / prep_threads; jobs[t].hi = n_in * (t + 1) / prep_threads;
    started[t] = n_in >= 1024 && pthread_create(&th[t], NULL, prep_main, &jobs[t]) == 0;
    if (!started[t]) prep_main(&jobs[t]);
  }
  for (int t = 0; t < prep_threads; ++t) if (started[t]) pthread_join(th[t], NULL);
  in->list = (ef_seq**)malloc((size_t)(2 * n_in + 1) * sizeof(ef_seq*));
  /* (no record is looked at here: two hundred thousand of them, fresh from other cores, are 0.01 s of misses --
   * what the scheduler wants to know of them, "does the next entry belong to this one?", is kept as a byte per entry) */
  in->has_rev = (unsigned char*)calloc((size_t)(2 * n_in + 1), 1);
  bool all = true;
  for (int t = 0; t < prep_threads; ++t) all = all && jobs[t].all_in_arena;
  for (long i = 0; i < n_in; ++i) {
    if (revs[i]) in->has_rev[in->n] = 1;
    in->list[in->n++] = ests[i];
    if (revs[i]) in->list[in->n++] = revs[i];
  }
  in->all_in_arena = all;
  free(ests); free(revs);
  return 0;
}

void ef_free_inputs(ef_inputs* in) {
  /* (two hundred thousand records that lie in the arena: looking at each one's flag is 200 000 misses, 0.02 s) */
  if (!in->all_in_arena) for (size_t k = 0; k < in->n; ++k) ef_seq_free(in->list[k]);
  free(in->list);
  free(in->has_rev); in->has_rev = NULL;
  ef_record_arena_free(in->arena); in->arena = NULL;
  ef_seq_free(in->gen);
  ef_genomic_epoch_bump();                   /* its address may come back with another gene behind it */
}

/* info-pid-<pid>.log (src/main-est-fact.c:106-115,181,221,233,243,290,333; log_info, src/util.c:222-268): one line per
 * phase boundary, "label <TAB> seconds since the epoch <TAB> the process's /proc/self/statm line".  The marks are taken
 * where the phases end in this program (reading and the GPU start-up run side by side here, so the times need not be
 * in the reference's order; the LINES are) and written with the other files. */
static struct { const char* label; unsigned long sec; char statm[96]; } info_marks[16];
static int n_info_marks;
static pthread_mutex_t info_mu = PTHREAD_MUTEX_INITIALIZER;
void ef_info_mark(const char* label) {
  struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
  char statm[96] = "NaN";
  FILE* f = fopen("/proc/self/statm", "r");
  if (f) { if (fgets(statm, sizeof statm, f)) { size_t n = strlen(statm); while (n && (statm[n - 1] == '\n' || statm[n - 1] == ' ')) statm[--n] = '\0'; } fclose(f); }
  pthread_mutex_lock(&info_mu);
  int k = 0;
  while (k < n_info_marks && strcmp(info_marks[k].label, label) != 0) ++k;       /* a later run of the process: the newest mark */
  if (k < 16) {
    info_marks[k].label = label; info_marks[k].sec = (unsigned long)ts.tv_sec;
    snprintf(info_marks[k].statm, sizeof info_marks[k].statm, "%s", statm);
    if (k == n_info_marks) ++n_info_marks;
  }
  pthread_mutex_unlock(&info_mu);
}
static void info_write(FILE* f, int upto_label) {
  static const char* order[7] = { "start", "data-io-end", "gst-construction-begin", "gst-preprocessing-begin",
                                  "gst-preprocessing-end", "est-processing-end", "end" };
  pthread_mutex_lock(&info_mu);
  for (int o = 0; o < upto_label; ++o)
    for (int k = 0; k < n_info_marks; ++k)
      if (!strcmp(info_marks[k].label, order[o])) fprintf(f, "%s\t%lu\t%s\n", order[o], info_marks[k].sec, info_marks[k].statm);
  pthread_mutex_unlock(&info_mu);
}

int ef_open_outputs(ef_outputs* o) {
  char buf[64];
  snprintf(buf, sizeof buf, "info-pid-%u.log", (unsigned)getpid());
  o->flog = fopen(buf, "w");
  memset(&o->fout, 0, sizeof o->fout); memset(&o->fests, 0, sizeof o->fests); memset(&o->fmeg, 0, sizeof o->fmeg);
  memset(&o->fpmeg, 0, sizeof o->fpmeg); memset(&o->ftmeg, 0, sizeof o->ftmeg); memset(&o->fintronic, 0, sizeof o->fintronic);
  o->fout.f = fopen("raw-multifasta-out.txt", "w");
  o->fmeg.f = fopen("megs.txt", "w");
  o->fpmeg.f = fopen("processed-megs.txt", "w");
  o->ftmeg.f = fopen("processed-megs-info.txt", "w");
  o->fests.f = fopen("processed-ests.txt", "w");
  o->fintronic.f = fopen("meg-edges.txt", "w");
  o->side.fmeg = &o->fmeg; o->side.fpmeg = &o->fpmeg; o->side.ftmeg = &o->ftmeg; o->side.fintronic = &o->fintronic;
  if (!o->flog || !o->fout.f || !o->fests.f || !o->fmeg.f || !o->fpmeg.f || !o->ftmeg.f || !o->fintronic.f) {
    fprintf(stderr, "* FATAL Cannot create an output file! Terminating\n");
    return 1;
  }
  return 0;
}

void ef_close_outputs(ef_outputs* o) {
  ef_info_mark("end");
  info_write(o->flog, 7);
  fclose(o->flog); fclose(o->fout.f); fclose(o->fests.f);
  fclose(o->fmeg.f); fclose(o->fpmeg.f); fclose(o->ftmeg.f); fclose(o->fintronic.f);
}

/* What the reference leaves on stderr when it ends (src/main-est-fact.c:321-335 + resource_usage_log,
 * src/util.c:184-208): its five timers, "End", user / system time and memory, in its own line format
 * (include/log.h: "* INFO (function        @ file:line ) message  ").  dist-scripts/pintron.py appends
 * est-fact's stderr to the pipeline log and people read these lines there.  The timers keep the
 * reference's names; what they cover here:
 *   Suffix Tree   GPU runtime start-up + construction (or load) of the device index
 *   Algorithm     pairings + MEGs of all sequences on the device (the prefetch stage's wall time)
 *   Compositions  the per-EST pipeline (embeddings, DP batches, refinement) on the worker threads
 *   IO            reading genomic.txt / ests.txt + writing the output files
 *   Total         process start to files on disk */
#include <sys/resource.h>
static void ref_info(const char* func, const char* file, int line, const char* msg) {
  char fn[17];
  const size_t l = strlen(func);
  for (size_t i = 0; i < 16; ++i) fn[i] = i < l ? func[i] : ' ';
  if (l + 2 >= 16) fn[15] = fn[14] = '.';
  fn[16] = '\0';
  const size_t fl = strlen(file);
  fprintf(stderr, "* INFO (%s@%20.20s:%-4d) %s  \n", fn, fl > 20 ? file + (fl - 20) : file, line, msg);
}
void ef_log_reference_timers(double suffix_tree_s, double algorithm_s, double compositions_s, double io_s, double total_s) {
  if (getenv("PINTRON_NO_TIMER_LOG")) return;
  static const char* names[5] = { "Suffix Tree", "Algorithm", "Compositions", "IO", "Total" };
  const double v[5] = { suffix_tree_s, algorithm_s, compositions_s, io_s, total_s };
  char msg[160];
  for (int k = 0; k < 5; ++k) {
    snprintf(msg, sizeof msg, "@Timer %-22s. Time elapsed: %15llu microsec", names[k], (unsigned long long)(v[k] > 0 ? v[k] * 1e6 : 0));
    ref_info("main", "src/main-est-fact.c", 321 + k, msg);
  }
  ref_info("main", "src/main-est-fact.c", 335, "End");
  struct rusage ru;
  if (getrusage(RUSAGE_SELF, &ru) == 0) {
    snprintf(msg, sizeof msg, "User time:   %10lds %7ldmicrosec.", (long)ru.ru_utime.tv_sec, (long)ru.ru_utime.tv_usec);
    ref_info("resource_usage_log", "src/util.c", 187, msg);
    snprintf(msg, sizeof msg, "System time: %10lds %7ldmicrosec.", (long)ru.ru_stime.tv_sec, (long)ru.ru_stime.tv_usec);
    ref_info("resource_usage_log", "src/util.c", 188, msg);
    unsigned size = 0;
    FILE* pf = fopen("/proc/self/statm", "r");
    if (pf && fscanf(pf, "%u", &size) == 1) snprintf(msg, sizeof msg, "Mem. used:   %10uKB", size);
    else snprintf(msg, sizeof msg, "Mem. used:         NaN KB");
    if (pf) fclose(pf);
    ref_info("resource_usage_log", "src/util.c", 199, msg);
  }
}

static double wall_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* main of est-fact (src/main-est-fact.c:90-339), one EST after the other, with the backend
 * supplied by the caller */
int ef_run(int argc, char** argv, ef_backend* (*open_backend)(const ef_seq* gen), void (*close_backend)(ef_backend*)) {
  ef_inputs in;
  const double t_start = wall_now();
  ef_info_mark("start");
  int rc = ef_load_inputs(argc, argv, &in);
  if (rc) return rc;
  ef_outputs out;
  if (ef_open_outputs(&out)) return 1;
  const double t_loaded = wall_now();
  ef_info_mark("data-io-end");
  ef_info_mark("gst-construction-begin");
  ef_backend* be = open_backend(in.gen);
  ef_info_mark("gst-preprocessing-begin");
  ef_info_mark("gst-preprocessing-end");
  if (!be) { fprintf(stderr, "* FATAL cannot initialise the compute backend (no MI355X / library)\n"); return 1; }
  const double t_index = wall_now();
  /* per-EST loop (:249-291) */
  bool reversed = false;
  for (size_t k = 0; k < in.n; ++k) {
    ef_seq* est = in.list[k];
    ef_est* fe = ef_compute_est_fact(in.gen, est, be, &in.cfg, &out.side);
    if (!efl_empty(fe->factorizations)) {
      ef_write_multifasta_output(in.gen, fe, &out.fout, in.cfg.retain_externals);
      ef_write_single_est_info(&out.fests, fe->info);
      if (!est->fixed_strand && !reversed) ++k;        /* skip the reverse-complement sibling */
      reversed = false;
    } else if (reversed || est->fixed_strand) {
      reversed = false;
    } else {
      reversed = true;
    }
    ef_est_free(fe);
  }
  const double t_done = wall_now();
  ef_info_mark("est-processing-end");
  close_backend(be);
  ef_close_outputs(&out);
  ef_free_inputs(&in);
  /* one EST after the other: pairings and compositions are not timed apart here */
  ef_log_reference_timers(t_index - t_loaded, 0.0, t_done - t_index, (t_loaded - t_start) + (wall_now() - t_done), wall_now() - t_start);
  return 0;
}
