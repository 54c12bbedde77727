/* est-fact options: defaults, command line, config.ini, config-dump.ini.
 * Option names, defaults and validity ranges are those of the reference's gengetopt schema
 * (src/options.ggo:48-370) and src/configuration.c:45-176; precedence: command line, then the
 * configuration file, then defaults (src/configuration.c:257-277). */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "estfact.h"

void ef_config_defaults(ef_config* c) {
  memset(c, 0, sizeof(*c));
  c->min_factor_len = 15; c->min_intron_length = 40; c->max_intron_length = 0;
  c->min_string_depth_rate = 0.2;
  c->max_prefix_discarded_rate = 0.60; c->max_suffix_discarded_rate = 0.60;
  c->max_prefix_discarded = 50; c->max_suffix_discarded = 50;
  c->max_site_difference = 50; c->max_number_of_factorizations = 0;
  c->max_coverage_diff = 0.05; c->max_exonNUM_diff = 5; c->max_gapLength_diff = 20;
  c->retain_externals = 1;
  c->max_pairings_in_MEG = 80; c->max_freq_shortest_pairing = 0.4;
  c->suffpref_length_for_intron = 70; c->suffpref_length_on_est = 30; c->suffpref_length_on_gen = 30;
  c->trans_red = true; c->short_edge_comp = true;
  c->max_single_factorization_time = 900;
  c->complexity_threshold = 20.0;
  strcpy(c->config_file, "config.ini");
}

typedef enum { T_INT, T_DBL, T_STR, T_FLAG, T_BOOLSTR } otype;
typedef struct { const char* name; char shortopt; otype type; size_t off; } optdef;
#define OFF(f) offsetof(ef_config, f)
/* the reference keeps three int-typed options in differently named fields */
typedef struct { int min_factor_length, min_distance_of_splice_sites, max_no_of_factorizations,
                 max_pairings_in_CMEG, max_single_factorization_time;
                 int no_trans_red, no_short_edge; } raw_ints;

static const char* const NAMES[] = {
  "config-file", "min-factor-length", "min-intron-length", "max-intron-length",
  "min-string-depth-rate", "max-prefix-discarded-rate", "max-suffix-discarded-rate",
  "max-prefix-discarded", "max-suffix-discarded", "min-distance-of-splice-sites",
  "max-no-of-factorizations", "max-difference-of-coverage", "max-difference-of-no-of-exons",
  "max-difference-of-gap-length", "complexity-threshold", "retain-externals",
  "max-pairings-in-CMEG", "max-shortest-pairing-frequence", "suff-pref-length-intron",
  "suff-pref-length-est", "suff-pref-length-genomic", "no-transitive-reduction",
  "no-short-edge-compaction", "max-single-factorization-time", NULL };
static const char SHORTS[] = { 'C', 'l', 'B', 0, 'd', 'p', 's', 'P', 'S', 'D', 0, 0, 0, 0, 0, 'E',
                               0, 0, 0, 0, 0, 0, 0, 0 };

/* applies option #k with text value v (NULL for flags); returns -1 on a bad value */
static int apply(ef_config* c, int k, const char* v) {
  char* end = NULL;
  const long iv = v ? strtol(v, &end, 10) : 0;
  const double dv = v ? strtod(v, NULL) : 0.0;
  switch (k) {
    case 0: if (!v) return -1; snprintf(c->config_file, sizeof c->config_file, "%s", v); return 0;
    case 1: if (iv <= 0) return -1; c->min_factor_len = (unsigned)iv; return 0;
    case 2: if (iv < 0) return -1; c->min_intron_length = (int)iv; return 0;
    case 3: if (iv < 0) return -1; c->max_intron_length = (int)iv; return 0;
    case 4: if (dv < 0.0 || dv > 1.0) return -1; c->min_string_depth_rate = dv; return 0;
    case 5: if (dv < 0.0 || dv > 1.0) return -1; c->max_prefix_discarded_rate = dv; return 0;
    case 6: if (dv < 0.0 || dv > 1.0) return -1; c->max_suffix_discarded_rate = dv; return 0;
    case 7: if (iv < 0) return -1; c->max_prefix_discarded = (int)iv; return 0;
    case 8: if (iv < 0) return -1; c->max_suffix_discarded = (int)iv; return 0;
    case 9: if (iv < 0) return -1; c->max_site_difference = (unsigned)iv; return 0;
    case 10: if (iv < 0) return -1; c->max_number_of_factorizations = (int)iv; return 0;
    case 11: if (dv < 0.0 || dv > 1.0) return -1; c->max_coverage_diff = dv; return 0;
    case 12: if (iv < -1) return -1; c->max_exonNUM_diff = (int)iv; return 0;
    case 13: if (iv < -1) return -1; c->max_gapLength_diff = (int)iv; return 0;
    case 14: if (dv <= 0.0) return -1; c->complexity_threshold = dv; return 0;
    case 15:
      if (v && !strcmp(v, "true")) c->retain_externals = 1;
      else if (v && !strcmp(v, "false")) c->retain_externals = 0;
      else return -1;
      return 0;
    case 16: if (iv < 0) return -1; c->max_pairings_in_MEG = (unsigned)iv; return 0;
    case 17: if (dv < 0.0 || dv > 1.0) return -1; c->max_freq_shortest_pairing = dv; return 0;
    case 18: if (iv <= 0) return -1; c->suffpref_length_for_intron = (int)iv; return 0;
    case 19: if (iv <= 0) return -1; c->suffpref_length_on_est = (int)iv; return 0;
    case 20: if (iv <= 0) return -1; c->suffpref_length_on_gen = (int)iv; return 0;
    case 21: c->trans_red = false; return 0;
    case 22: c->short_edge_comp = false; return 0;
    case 23: if (iv < 0) return -1; c->max_single_factorization_time = (unsigned)iv; return 0;
  }
  return -1;
}

static bool is_flag(int k) { return k == 21 || k == 22; }

static int find_long(const char* name, size_t len) {
  for (int k = 0; NAMES[k]; ++k)
    if (strlen(NAMES[k]) == len && !strncmp(NAMES[k], name, len)) return k;
  return -1;
}

static int parse_file(ef_config* c, const char* path, const bool* given) {
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  char line[4096];
  while (fgets(line, sizeof line, f)) {
    char* p = line;
    while (*p == ' ' || *p == '\t') ++p;
    if (*p == '#' || *p == '\n' || *p == '\0') continue;
    char* e = p;
    while (*e && *e != '=' && *e != ' ' && *e != '\t' && *e != '\n') ++e;
    const int k = find_long(p, (size_t)(e - p));
    if (k < 0) { fprintf(stderr, "est-fact: unknown option '%.*s' in %s\n", (int)(e - p), p, path); fclose(f); return -1; }
    while (*e == ' ' || *e == '\t' || *e == '=') ++e;
    char* v = e;
    size_t vl = strlen(v);
    while (vl && (v[vl - 1] == '\n' || v[vl - 1] == '\r' || v[vl - 1] == ' ' || v[vl - 1] == '\t')) v[--vl] = '\0';
    if (vl >= 2 && v[0] == '"' && v[vl - 1] == '"') { v[vl - 1] = '\0'; ++v; }
    if (given[k]) continue;                       /* the command line wins */
    if (apply(c, k, is_flag(k) ? NULL : v) != 0) { fprintf(stderr, "est-fact: invalid value for '%s' in %s\n", NAMES[k], path); fclose(f); return -1; }
  }
  fclose(f);
  return 0;
}

static void dump(const ef_config* c) {                 /* cmdline_parser_file_save format */
  FILE* f = fopen("./config-dump.ini", "w");
  if (!f) return;
  fprintf(f, "# This file was written by est-fact (MI355X build)\n");
  fprintf(f, "config-file=\"%s\"\n", c->config_file);
  fprintf(f, "min-factor-length=\"%u\"\nmin-intron-length=\"%d\"\nmax-intron-length=\"%d\"\n",
          c->min_factor_len, c->min_intron_length, c->max_intron_length);
  fprintf(f, "min-string-depth-rate=\"%g\"\nmax-prefix-discarded-rate=\"%g\"\nmax-suffix-discarded-rate=\"%g\"\n",
          c->min_string_depth_rate, c->max_prefix_discarded_rate, c->max_suffix_discarded_rate);
  fprintf(f, "max-prefix-discarded=\"%d\"\nmax-suffix-discarded=\"%d\"\nmin-distance-of-splice-sites=\"%u\"\n",
          c->max_prefix_discarded, c->max_suffix_discarded, c->max_site_difference);
  fprintf(f, "max-no-of-factorizations=\"%d\"\nmax-difference-of-coverage=\"%g\"\nmax-difference-of-no-of-exons=\"%d\"\n",
          c->max_number_of_factorizations, c->max_coverage_diff, c->max_exonNUM_diff);
  fprintf(f, "max-difference-of-gap-length=\"%d\"\ncomplexity-threshold=\"%g\"\nretain-externals=\"%s\"\n",
          c->max_gapLength_diff, c->complexity_threshold, c->retain_externals ? "true" : "false");
  fprintf(f, "max-pairings-in-CMEG=\"%u\"\nmax-shortest-pairing-frequence=\"%g\"\n",
          c->max_pairings_in_MEG, c->max_freq_shortest_pairing);
  fprintf(f, "suff-pref-length-intron=\"%d\"\nsuff-pref-length-est=\"%d\"\nsuff-pref-length-genomic=\"%d\"\n",
          c->suffpref_length_for_intron, c->suffpref_length_on_est, c->suffpref_length_on_gen);
  if (!c->trans_red) fprintf(f, "no-transitive-reduction\n");
  if (!c->short_edge_comp) fprintf(f, "no-short-edge-compaction\n");
  fprintf(f, "max-single-factorization-time=\"%u\"\n", c->max_single_factorization_time);
  fclose(f);
}

int ef_ahead_on = 1;
int ef_chain_fast_path = 1;
int ef_endpoint_checks = 1;
int ef_prof_on;                                   /* PINTRON_PROFILE, read once here */
_Thread_local ef_prof_state ef_prof;
unsigned long long ef_work_budget_override;     /* PINTRON_WORK_BUDGET, read once here (single-threaded); used by ef_fact.c */

int ef_config_load(ef_config* c, int argc, char** argv) {
  ef_config_defaults(c);
  { const char* e = getenv("PINTRON_ENDPOINT_CHECKS"); ef_endpoint_checks = !(e && e[0] == '0'); }
  { const char* e = getenv("PINTRON_CHAIN"); ef_chain_fast_path = !(e && e[0] == '0' && e[1] == '\0'); }
  { const char* e = getenv("PINTRON_AHEAD"); ef_ahead_on = !(e && e[0] == '0' && e[1] == '\0'); }
  { const char* e = getenv("PINTRON_PROFILE"); ef_prof_on = e && e[0] && !(e[0] == '0' && e[1] == '\0'); }
  { const char* e = getenv("PINTRON_WORK_BUDGET"); ef_work_budget_override = e && atoll(e) > 0 ? (unsigned long long)atoll(e) : 0ull; }
  bool given[32] = { false };
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    int k = -1;
    const char* v = NULL;
    if (a[0] == '-' && a[1] == '-') {
      const char* eq = strchr(a + 2, '=');
      k = find_long(a + 2, eq ? (size_t)(eq - a - 2) : strlen(a + 2));
      if (eq) v = eq + 1;
    } else if (a[0] == '-' && a[1] && strchr("ClBdpsPSDE", a[1])) {
      for (int q = 0; NAMES[q]; ++q) if (SHORTS[q] == a[1]) k = q;
      if (a[2]) v = a + 2;
    }
    if (k < 0) { fprintf(stderr, "est-fact: unrecognized option '%s'\n", a); return -1; }
    if (!is_flag(k) && !v) {
      if (i + 1 >= argc) { fprintf(stderr, "est-fact: option '%s' requires an argument\n", a); return -1; }
      v = argv[++i];
    }
    if (apply(c, k, is_flag(k) ? NULL : v) != 0) { fprintf(stderr, "est-fact: invalid argument for option '%s'\n", NAMES[k]); return -1; }
    given[k] = true;
  }
  if (access(c->config_file, R_OK) == 0 && parse_file(c, c->config_file, given) != 0) return -1;
  dump(c);
  return 0;
}
