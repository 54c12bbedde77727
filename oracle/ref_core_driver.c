/*
 * TEST INFRASTRUCTURE ONLY (oracle/): never linked into or called by the product.
 *
 * `est-fact-core` = this file + oracle/_ref/libpintron_ref_core.so.
 *
 * libpintron_ref_core.so holds ONLY reference object code, compiled from the sources where they
 * lie under /root/reference: every algorithmic file of the est-fact path (suffix tree, pairings,
 * MEG, embeddings, every DP, refinement, classification, readers and writers).  Three reference
 * files are NOT in it, and nothing of ours stands in for them:
 *   src/configuration.c   includes the gengetopt-generated options.h (reference Makefile:579-583);
 *                         gengetopt is not in the image => unbuildable here;
 *   src/main-est-fact.c   needs config_create() of configuration.c;
 *   src/compute-est-fact.c needs config_clone()/config_destroy() of configuration.c.
 *
 * What this file is: OUR CPU restatement of the control flow of those two driver files --
 * main-est-fact.c:70-339 (input preparation, the per-EST loop with the reverse-complement
 * sibling rule, the six output files) and compute-est-fact.c:90-293 (the retry loops over
 * inc_pairing_len around MEG construction and factorization) -- calling the reference's own
 * object code for everything they call.  It fills the public `struct _configuration`
 * (include/configuration.h:39-135) itself with the defaults of src/options.ggo:94-370; no option
 * or config.ini parsing exists here (the pipeline runs est-fact without options,
 * dist-scripts/pintron.py:878-884).  Overrides for tests: PINTRON_REF_MIN_FACTOR_LEN.
 *
 * How the restated control flow is pinned: tools/pin_regression.py pipes the output of this
 * program through the reference's unmodified min-factorization and intron-agreement (both
 * compiled here without any file of ours) and compares the predicted introns and their
 * supporting-EST factor boundaries with the reference-held regressionTest/ goldens
 * (referenceOutput/full.json), the way regressionTest/testPIntronOutput.c:116-224 does.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "types.h"
#include "list.h"
#include "ext_array.h"
#include "util.h"
#include "my_time.h"
#include "configuration.h"
#include "io-multifasta.h"
#include "io-meg.h"
#include "aug_suffix_tree.h"
#include "max-emb-graph.h"
#include "meg-simplification.h"
#include "est-factorizations.h"
#include "factorization-refinement.h"

struct outputs {
  FILE* raw;        /* raw-multifasta-out.txt */
  FILE* ests;       /* processed-ests.txt */
  FILE* megs;       /* megs.txt */
  FILE* pmegs;      /* processed-megs.txt */
  FILE* pmegs_info; /* processed-megs-info.txt */
  FILE* edges;      /* meg-edges.txt */
};

static FILE* must_open(const char* name, const char* mode) {
  FILE* f = fopen(name, mode);
  if (!f) { fprintf(stderr, "est-fact-core: cannot open %s\n", name); exit(2); }
  return f;
}

/* options.ggo:94-370 */
void ref_core_default_config(struct _configuration* c) {
  memset(c, 0, sizeof *c);
  c->min_factor_len = 15;
  c->min_intron_length = 40;
  c->max_intron_length = 0;
  c->min_string_depth_rate = 0.2;
  c->max_prefix_discarded_rate = 0.60;
  c->max_suffix_discarded_rate = 0.60;
  c->max_prefix_discarded = 50;
  c->max_suffix_discarded = 50;
  c->max_site_difference = 50;
  c->max_number_of_factorizations = 0;
  c->max_coverage_diff = 0.05;
  c->max_exonNUM_diff = 5;
  c->max_gapLength_diff = 20;
  c->complexity_threshold = 20.0;
  c->retain_externals = 1;
  c->max_pairings_in_MEG = 80;
  c->max_freq_shortest_pairing = 0.4;
  c->suffpref_length_for_intron = 70;
  c->suffpref_length_on_est = 30;
  c->suffpref_length_on_gen = 30;
  c->trans_red = true;          /* "no-transitive-reduction" flag off */
  c->short_edge_comp = true;    /* "no-short-edge-compaction" flag off */
  c->max_single_factorization_time = 900;
}

/* TEST INFRASTRUCTURE: the options of the run, for the tests that pin est-fact under non-default options.
 * configuration.c (which needs the gengetopt-generated options.h) is not built here, so what it does with the
 * parsed options is restated: `ref-options.ini` in the current directory (or the file PINTRON_REF_OPTIONS names),
 * in the format the reference itself saves its configuration in (config-dump.ini, written by
 * cmdline_parser_file_save: `long-option-name="value"`, a flag as its bare name), is applied over the defaults
 * with the checks of check_and_copy (src/configuration.c:45-176); a value it would refuse ends the program with
 * status 3.  The option -> field mapping is check_and_copy's. */
static void bad_option(const char* name, const char* value) {
  fprintf(stderr, "est-fact-core: option '%s' refused (value '%s')\n", name, value ? value : "");
  exit(3);
}
static void ref_core_config_from_file(struct _configuration* c, const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return;
  char line[1024];
  while (fgets(line, sizeof line, f)) {
    char* p = line;
    while (*p == ' ' || *p == '\t') ++p;
    if (*p == '#' || *p == '\n' || *p == '\0') continue;
    char* e = p;
    while (*e && *e != '=' && *e != ' ' && *e != '\t' && *e != '\n') ++e;
    char name[128];
    snprintf(name, sizeof name, "%.*s", (int)(e - p), p);
    while (*e == ' ' || *e == '\t' || *e == '=') ++e;
    char* v = e;
    size_t vl = strlen(v);
    while (vl && (v[vl - 1] == '\n' || v[vl - 1] == '\r' || v[vl - 1] == ' ' || v[vl - 1] == '\t')) v[--vl] = '\0';
    if (vl >= 2 && v[0] == '"' && v[vl - 1] == '"') { v[vl - 1] = '\0'; ++v; }
    const long iv = strtol(v, NULL, 10);
    const double dv = strtod(v, NULL);
#define OPT(n) (strcmp(name, n) == 0)
    if (OPT("config-file")) continue;
    else if (OPT("min-factor-length")) { if (iv <= 0) bad_option(name, v); c->min_factor_len = (unsigned)iv; }
    else if (OPT("min-intron-length")) { if (iv < 0) bad_option(name, v); c->min_intron_length = (int)iv; }
    else if (OPT("max-intron-length")) { if (iv < 0) bad_option(name, v); c->max_intron_length = (int)iv; }
    else if (OPT("min-string-depth-rate")) { if (dv < 0.0 || dv > 1.0) bad_option(name, v); c->min_string_depth_rate = dv; }
    else if (OPT("max-prefix-discarded-rate")) { if (dv < 0.0 || dv > 1.0) bad_option(name, v); c->max_prefix_discarded_rate = dv; }
    else if (OPT("max-suffix-discarded-rate")) { if (dv < 0.0 || dv > 1.0) bad_option(name, v); c->max_suffix_discarded_rate = dv; }
    else if (OPT("max-prefix-discarded")) { if (iv < 0) bad_option(name, v); c->max_prefix_discarded = (int)iv; }
    else if (OPT("max-suffix-discarded")) { if (iv < 0) bad_option(name, v); c->max_suffix_discarded = (int)iv; }
    else if (OPT("min-distance-of-splice-sites")) { if (iv < 0) bad_option(name, v); c->max_site_difference = (unsigned)iv; }
    else if (OPT("max-no-of-factorizations")) { if (iv < 0) bad_option(name, v); c->max_number_of_factorizations = (int)iv; }
    else if (OPT("max-difference-of-coverage")) { if (dv < 0.0 || dv > 1.0) bad_option(name, v); c->max_coverage_diff = dv; }
    else if (OPT("max-difference-of-no-of-exons")) { if (iv < -1) bad_option(name, v); c->max_exonNUM_diff = (int)iv; }
    else if (OPT("max-difference-of-gap-length")) { if (iv < -1) bad_option(name, v); c->max_gapLength_diff = (int)iv; }
    else if (OPT("complexity-threshold")) { if (dv <= 0.0) bad_option(name, v); c->complexity_threshold = dv; }
    else if (OPT("retain-externals")) {
      if (!strcmp(v, "true")) c->retain_externals = 1; else if (!strcmp(v, "false")) c->retain_externals = 0; else bad_option(name, v);
    }
    else if (OPT("max-pairings-in-CMEG")) { if (iv < 0) bad_option(name, v); c->max_pairings_in_MEG = (unsigned)iv; }
    else if (OPT("max-shortest-pairing-frequence")) { if (dv < 0.0 || dv > 1.0) bad_option(name, v); c->max_freq_shortest_pairing = dv; }
    else if (OPT("suff-pref-length-intron")) { if (iv <= 0) bad_option(name, v); c->suffpref_length_for_intron = (int)iv; }
    else if (OPT("suff-pref-length-est")) { if (iv <= 0) bad_option(name, v); c->suffpref_length_on_est = (int)iv; }
    else if (OPT("suff-pref-length-genomic")) { if (iv <= 0) bad_option(name, v); c->suffpref_length_on_gen = (int)iv; }
    else if (OPT("no-transitive-reduction")) c->trans_red = false;
    else if (OPT("no-short-edge-compaction")) c->short_edge_comp = false;
    else if (OPT("max-single-factorization-time")) { if (iv < 0) bad_option(name, v); c->max_single_factorization_time = (unsigned)iv; }
    else { fprintf(stderr, "est-fact-core: unknown option '%s' in %s\n", name, path); exit(3); }
#undef OPT
  }
  fclose(f);
}

static char* dup_or_null(const char* s) { return s ? alloc_and_copy(s) : NULL; }

/* main-est-fact.c:70-88: the reverse-complement sibling of a sequence whose strand is not fixed;
 * the polyA/polyT bookkeeping swaps ends and letters. */
static pEST_info sibling_of(pEST_info est) {
  pEST_info rc = EST_info_create();
  rc->EST_seq = dup_or_null(est->EST_seq);
  rc->original_EST_seq = dup_or_null(est->original_EST_seq);
  reverse_and_complement(rc);
  rc->EST_id = dup_or_null(est->EST_id);
  rc->EST_gb = dup_or_null(est->EST_gb);
  rc->EST_chr = dup_or_null(est->EST_chr);
  rc->EST_strand_as_read = dup_or_null(est->EST_strand_as_read);
  rc->EST_strand = -est->EST_strand;
  rc->fixed_strand = est->fixed_strand;
  rc->pref_polyA_length = est->suff_polyT_length;
  rc->suff_polyA_length = est->pref_polyT_length;
  rc->pref_polyT_length = est->suff_polyA_length;
  rc->suff_polyT_length = est->pref_polyA_length;
  return rc;
}

/* compute-est-fact.c:90-146: vertex set at min_factor_len + inc, edges, clean-up; a MEG judged too
 * complex is rebuilt with a longer minimum factor while the pattern leaves room for it. */
static pext_array meg_for(pEST_info est, LST_STree* tree, ppreproc_gen pg,
                          const struct _configuration* shared, size_t* inc, pmytime t_meg) {
  struct _configuration cfg = *shared;
  for (;;) {
    cfg.min_factor_len = shared->min_factor_len + (unsigned)*inc;
    pext_array V = build_vertex_set(est, tree, pg, &cfg);
    MYTIME_reset(t_meg);
    MYTIME_start(t_meg);
    build_edge_set(V, &cfg);
    simplify_meg(V, &cfg);
    if (cfg.trans_red) {
      pgraph g = meg2graph(V);
      transitive_reduction(g);
      graph_destroy(g);
    }
    bool complex = is_too_complex_for_compaction(V, &cfg);
    if (!complex && cfg.short_edge_comp) compact_short_edges(V, &cfg);
    if (!complex) complex = is_too_complex(V, &cfg);
    /* is_too_complex() sees the raised min_factor_len; the room test uses the shared one */
    const bool room = shared->min_factor_len + *inc + 1 + 2 < EA_size(V);
    MYTIME_stop(t_meg);
    if (!complex || !room) return V;
    ++*inc;
    EA_destroy(V, (delete_function)vi_destroy);
  }
}

/* compute-est-fact.c:192-293 */
static pEST factorize_one(pEST_info gen, pEST_info est, LST_STree* tree, ppreproc_gen pg,
                          const struct _configuration* shared, struct outputs* out) {
  pmytime t_meg = MYTIME_create_with_name("MEGs");
  pmytime t_fact = MYTIME_create_with_name("Internal Comp.");
  struct _configuration cfg = *shared;
  size_t inc = 0, prev_pairings = 0, prev_edges = 0;
  pEST result = NULL;
  bool again;
  do {
    pext_array V;
    size_t n_pairings, n_edges;
    for (;;) {       /* a retry must shrink the MEG, else lengthen the factor again (:225-243) */
      V = meg_for(est, tree, pg, shared, &inc, t_meg);
      MEG_stats(V, &n_pairings, &n_edges);
      if (!(prev_pairings > 2 && prev_edges > 0 &&
            (prev_pairings <= n_pairings || prev_edges <= n_edges)))
        break;
      ++inc;
      EA_destroy(V, (delete_function)vi_destroy);
    }
    prev_pairings = n_pairings;
    prev_edges = n_edges;

    /* compute-est-fact.c:154-190 */
    pmytime_timeout limit = MYTIME_timeout_create(cfg.max_single_factorization_time);
    MYTIME_reset(t_fact);
    MYTIME_start(t_fact);
    result = get_EST_factorizations(est, V, &cfg, gen, limit);
    bool expired = MYTIME_timeout_expired(limit);
    if (result != NULL) {
      refine_EST_factorizations(gen, result, &cfg);
      remove_factorizations_with_very_small_exons(result->factorizations);
      if (!list_is_empty(result->factorizations))
        remove_duplicated_factorizations(result->factorizations);
    }
    MYTIME_stop(t_fact);
    expired = expired || MYTIME_timeout_expired(limit);
    MYTIME_timeout_destroy(limit);

    const bool aligned = result != NULL && !list_is_empty(result->factorizations);
    if (!expired || aligned) {        /* report_meg, :73-88 */
      fprintf(out->megs, "\n\n***********\n\n");
      write_single_EST_info(out->megs, est);
      meg_write(out->megs, V);
      fflush(out->megs);
    }
    again = false;
    if (aligned) {
      fprintf(out->edges, ">%s\n", est->EST_id);
      add_intronic_edges_to_file(out->edges, V);
      write_single_EST_info(out->pmegs, est);
      meg_write(out->pmegs, V);
      fprintf(out->pmegs_info, "%llu %llu %zu\n", MYTIME_getinterval(t_meg),
              MYTIME_getinterval(t_fact), list_size(result->factorizations));
    } else if (expired) {
      ++inc;
      again = true;
    }
    EA_destroy(V, (delete_function)vi_destroy);
  } while (again);
  MYTIME_destroy(t_meg);
  MYTIME_destroy(t_fact);
  return result;
}

int main(void) {
  struct _configuration cfg;
  ref_core_default_config(&cfg);
  const char* l = getenv("PINTRON_REF_MIN_FACTOR_LEN");
  if (l && atoi(l) > 0) cfg.min_factor_len = (unsigned)atoi(l);
  ref_core_config_from_file(&cfg, getenv("PINTRON_REF_OPTIONS") ? getenv("PINTRON_REF_OPTIONS") : "ref-options.ini");

  /* main-est-fact.c:116-135 */
  FILE* fgen = must_open("genomic.txt", "r");
  plist gens = read_multifasta(fgen);
  fclose(fgen);
  if (list_size(gens) != 1) { fprintf(stderr, "est-fact-core: genomic.txt must hold one record\n"); return 2; }
  pEST_info gen = (pEST_info)list_head(gens);
  list_destroy(gens, noop_free);
  parse_genomic_header(gen);
  Ntails_removal(gen);

  FILE* fests = must_open("ests.txt", "r");
  plist read = read_multifasta(fests);
  fclose(fests);

  struct outputs out;
  out.raw = must_open("raw-multifasta-out.txt", "w");
  out.megs = must_open("megs.txt", "w");
  out.pmegs = must_open("processed-megs.txt", "w");
  out.pmegs_info = must_open("processed-megs-info.txt", "w");
  out.ests = must_open("processed-ests.txt", "w");
  out.edges = must_open("meg-edges.txt", "w");

  /* main-est-fact.c:248-273: every sequence, followed by its reverse complement unless the
   * strand is fixed */
  plist work = list_create();
  plistit it = list_first(read);
  while (listit_has_next(it)) {
    pEST_info est = (pEST_info)listit_next(it);
    set_EST_GB_identification(est);
    set_EST_Strand_and_RC(est, gen);
    list_add_to_tail(work, est);
    polyAT_substitution(est);
    if (!est->fixed_strand) {
      pEST_info rc = sibling_of(est);
      list_add_to_tail(work, rc);
      polyAT_substitution(rc);
    }
  }
  listit_destroy(it);
  list_destroy(read, (delete_function)noop_free);

  /* main-est-fact.c:280-300 */
  LST_StringSet* set = lst_stringset_new();
  LST_String* text = PALLOC(LST_String);
  lst_string_init(text, gen->EST_seq, sizeof(char), strlen(gen->EST_seq));
  lst_stringset_add(set, text);
  LST_STree* tree = lst_stree_new(set);
  ppreproc_gen pg = PGen_create();
  preprocess_text(gen, pg);
  stree_preprocess(tree, pg, &cfg);

  /* main-est-fact.c:302-351 */
  bool on_sibling = false;
  it = list_first(work);
  while (listit_has_next(it)) {
    pEST_info est = (pEST_info)listit_next(it);
    pEST fe = factorize_one(gen, est, tree, pg, &cfg, &out);
    if (!list_is_empty(fe->factorizations)) {
      write_multifasta_output(gen, fe, out.raw, cfg.retain_externals);
      write_single_EST_info(out.ests, fe->info);
      if (!est->fixed_strand && !on_sibling) listit_next(it);   /* skip its sibling */
      on_sibling = false;
    } else {
      on_sibling = !(on_sibling || est->fixed_strand);
    }
    EST_destroy_just_factorizations(fe);
  }
  listit_destroy(it);

  fclose(out.raw); fclose(out.ests); fclose(out.megs);
  fclose(out.pmegs); fclose(out.pmegs_info); fclose(out.edges);
  return 0;
}
