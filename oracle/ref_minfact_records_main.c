/*
 * TEST INFRASTRUCTURE + reference-side binding shown in INTEGRATION.md section 6 (never part of the
 * product): the reference's min-factorization with a front end that reads est-fact's PACKED
 * RECORDS (include/pintron_records.h) instead of parsing raw-multifasta-out.txt.
 *
 *   min-factorization-records <records.bin> <processed-ests.txt>   > out-agree.txt
 *
 * `factorizations_from_records` builds exactly the structure read_factorizations returns
 * (src/io-factorizations.c:194-238: one pEST per run of records with the same header, its
 * factorizations in file order, addFactorization :104-166 for the exon fields and the two flags);
 * the headers come from processed-ests.txt, which lists the aligned ESTs in the same order.
 * main() is the flow of src/main-min-factorization.c:39-187 with that one call replaced; every
 * routine it calls is the reference's own object code, compiled from where it lies (oracle/Makefile).
 * tests/ compare its out-agree.txt with the unmodified min-factorization fed the text file.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "types.h"
#include "list.h"
#include "bool_list.h"
#include "util.h"
#include "min_factorization.h"
#include "bit_vector.h"
#include "color_matrix.h"
#include "simplify_matrix.h"

#include "../include/pintron_records.h"

static char* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
  char* b = (char*)malloc((size_t)n + 1);
  if (n > 0 && fread(b, 1, (size_t)n, f) != (size_t)n) { perror(path); exit(2); }
  b[n] = '\0'; fclose(f);
  *len = (size_t)n;
  return b;
}

plist factorizations_from_records(const void* records, size_t len, const char* processed_ests) {
  plist ests = list_create();
  pfr_reader r; pfr_open(&r, records, len);
  pfr_est e; pfr_factorization f;
  const char* ep = processed_ests;
  pEST cur = NULL;
  int rc;
  while ((rc = pfr_next_est(&r, &e)) == 1) {
    /* processed-ests.txt: ">header\n<sequence>\n" per aligned EST, same order as the records */
    const char* h = ep; while (*ep && *ep != '\n') ++ep;
    const size_t hl = (size_t)(ep - h); if (*ep) ++ep;
    while (*ep && *ep != '\n') ++ep;
    if (*ep) ++ep;
    if (hl == 0 || h[0] != '>') { fprintf(stderr, "processed-ests.txt does not match the records\n"); exit(1); }
    if (e.n_factorizations == 0) continue;      /* an aligned EST with nothing to print (--retain-externals=false): not in the text either */
    char* id = (char*)malloc(hl); memcpy(id, h + 1, hl - 1); id[hl - 1] = '\0';
    /* read_factorizations starts a new EST when the header differs from the previous one */
    if (cur == NULL || strcmp(cur->info->EST_id, id) != 0) {
      cur = EST_create();
      cur->info = EST_info_create();
      cur->info->EST_id = id;
      cur->factorizations = list_create();
      cur->polyA_signals = boollist_create();
      cur->polyadenil_signals = boollist_create();
      list_add_to_tail(ests, cur);
    } else free(id);
    while ((rc = pfr_next_factorization(&r, &f)) == 1) {
      plist fact = list_create();
      for (uint16_t k = 0; k < f.n_exons; ++k) {
        const pfr_exon x = pfr_exon_at(&f, k);
        pfactor ft = factor_create();
        /* the zero clamps of addFactorization (:140-143) */
        ft->EST_start = x.est_start == 0 ? 1 : x.est_start;
        ft->EST_end = x.est_end == 0 ? 1 : x.est_end;
        ft->GEN_start = x.est_start == 0 ? 1 : x.gen_start;
        ft->GEN_end = x.est_start == 0 ? 1 : x.gen_end;
        list_add_to_tail(fact, ft);
      }
      list_add_to_tail(cur->factorizations, fact);
      boollist_add_to_tail(cur->polyA_signals, f.polya == 1);
      boollist_add_to_tail(cur->polyadenil_signals, f.polyad == 1);
    }
    if (rc < 0) break;
  }
  if (rc < 0) { fprintf(stderr, "records: truncated or inconsistent\n"); exit(1); }
  return ests;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s records.bin processed-ests.txt > out-agree.txt\n", argv[0]); return 2; }
  size_t rl, el;
  char* rec = slurp(argv[1], &rl);
  char* pe = slurp(argv[2], &el);
  plist p = factorizations_from_records(rec, rl, pe);
  /* src/main-min-factorization.c:47-156 from here on */
  pbit_vect bv = NULL;
  plist unique_factors = color_matrix_create(p, false);
  psimpl psimp = simplification(p, unique_factors);
  psimpl_print(psimp);
  plist pl = color_matrix_simplified_create(p, psimp);
  if (!BV_all_true(psimp->ests_ok)) bv = min_fact(pl);
  print_factorizations_result(bv, p, unique_factors, psimp);
  free(rec); free(pe);
  return 0;
}
