/*
 * TEST INFRASTRUCTURE ONLY (oracle/): linked into oracle/_ref/libpintron_ref.so, never into the product.
 *
 * The reference's src/configuration.c cannot be compiled in this image: it includes the
 * gengetopt-generated options.h (reference Makefile:579-583), and gengetopt is absent.  This file
 * is NOT a stand-in for that generated parser: it implements the three functions of the public
 * header include/configuration.h over `struct _configuration` (include/configuration.h:39-135),
 * returning the defaults documented in src/options.ggo:94-370.  Command-line options and
 * config.ini are NOT parsed; the only override is the environment variable
 * PINTRON_REF_MIN_FACTOR_LEN (used by tests to exercise a non-default `-l`).
 * The reference pipeline calls est-fact without options (dist-scripts/pintron.py:878-884).
 */
#include <stdlib.h>
#include <string.h>
#include "configuration.h"

pconfiguration ref_default_config(void) {
  pconfiguration c = (pconfiguration)calloc(1, sizeof(struct _configuration));
  c->min_factor_len = 15;               /* options.ggo:94-101  */
  c->min_intron_length = 40;            /* :104-111 */
  c->max_intron_length = 0;             /* :114-121 */
  c->min_string_depth_rate = 0.2;       /* :124-137 */
  c->max_prefix_discarded_rate = 0.60;  /* :140-149 */
  c->max_suffix_discarded_rate = 0.60;  /* :152-161 */
  c->max_prefix_discarded = 50;         /* :164-173 */
  c->max_suffix_discarded = 50;         /* :176-185 */
  c->max_site_difference = 50;          /* :188-197 */
  c->max_number_of_factorizations = 0;  /* :200-208 */
  c->max_coverage_diff = 0.05;          /* :211-221 */
  c->max_exonNUM_diff = 5;              /* :224-235 */
  c->max_gapLength_diff = 20;           /* :238-249 */
  c->complexity_threshold = 20.0;       /* :252-261 */
  c->retain_externals = 1;              /* :264-272 */
  c->max_pairings_in_MEG = 80;          /* :275-289 */
  c->max_freq_shortest_pairing = 0.4;   /* :293-308 */
  c->suffpref_length_for_intron = 70;   /* :311-320 */
  c->suffpref_length_on_est = 30;       /* :323-332 */
  c->suffpref_length_on_gen = 30;       /* :335-344 */
  c->trans_red = true;                  /* :347-349 (flag off => reduction performed) */
  c->short_edge_comp = true;            /* :351-353 */
  c->max_single_factorization_time = 900; /* :362-369 */
  const char* l = getenv("PINTRON_REF_MIN_FACTOR_LEN");
  if (l && atoi(l) > 0) c->min_factor_len = (unsigned)atoi(l);
  return c;
}

pconfiguration config_create(int argc, char** argv) {
  (void)argc; (void)argv;
  return ref_default_config();
}

pconfiguration config_clone(pconfiguration src) {
  pconfiguration c = (pconfiguration)malloc(sizeof(struct _configuration));
  memcpy(c, src, sizeof(struct _configuration));
  return c;
}

void config_destroy(pconfiguration config) { free(config); }
