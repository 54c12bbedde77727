/* Test consumer of include/pintron_records.h: rebuilds raw-multifasta-out.txt from the packed
 * records, processed-ests.txt (header + strand-corrected sequence of every aligned EST, same order)
 * and genomic.txt -- the C twin of pintron_amd.estfact.format_raw_multifasta.
 *   records_to_text <records.bin> <processed-ests.txt> <genomic.txt>  > raw-multifasta-out.txt */
#include <stdio.h>
#include <stdlib.h>
#include "../../include/pintron_records.h"

static char* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
  char* b = (char*)malloc((size_t)n + 1);
  if (fread(b, 1, (size_t)n, f) != (size_t)n) { perror(path); exit(2); }
  b[n] = '\0'; fclose(f);
  *len = (size_t)n;
  return b;
}

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s records.bin processed-ests.txt genomic.txt\n", argv[0]); return 2; }
  size_t rl, el, gl;
  char* rec = slurp(argv[1], &rl); char* ests = slurp(argv[2], &el); char* gen = slurp(argv[3], &gl);
  /* genomic sequence: every line after the header, joined */
  char* g = (char*)malloc(gl + 1); size_t gn = 0;
  { const char* p = gen; while (*p && *p != '\n') ++p; for (; *p; ++p) if (*p != '\n' && *p != '\r') g[gn++] = *p; }
  pfr_reader r; pfr_open(&r, rec, rl);
  pfr_est e; pfr_factorization f;
  const char* ep = ests;
  int rc;
  while ((rc = pfr_next_est(&r, &e)) == 1) {
    /* the next two lines of processed-ests.txt: ">header" and the sequence */
    const char* h = ep; while (*ep && *ep != '\n') ++ep;
    const size_t hl = (size_t)(ep - h); if (*ep) ++ep;
    const char* s = ep; while (*ep && *ep != '\n') ++ep;
    if (*ep) ++ep;
    while ((rc = pfr_next_factorization(&r, &f)) == 1) {
      printf("%.*s\n#polya=%d\n#polyad=%d\n", (int)hl, h, f.polya, f.polyad);
      for (uint16_t k = 0; k < f.n_exons; ++k) {
        const pfr_exon x = pfr_exon_at(&f, k);
        printf("%d %d %d %d %.*s %.*s\n", x.est_start, x.est_end, x.gen_start, x.gen_end,
               x.est_end - x.est_start + 1, s + x.est_start - 1, x.gen_end - x.gen_start + 1, g + x.gen_start - 1);
      }
    }
    if (rc < 0) break;
  }
  if (rc < 0) { fprintf(stderr, "records: truncated or inconsistent\n"); return 1; }
  return 0;
}
