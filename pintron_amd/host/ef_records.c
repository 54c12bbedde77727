/* raw-multifasta-out.txt rebuilt from the packed factorization records (include/pintron_records.h).
 *
 * The text est-fact prints per factorization (write_multifasta_output, src/io-multifasta.c:187-229) is the
 * header of the EST, the two polyA flags and, per exon, four numbers followed by the exon's stretch of the EST
 * and of the genomic sequence: everything but the numbers and flags is a substring of processed-ests.txt
 * (header + strand-corrected sequence of every aligned EST, src/io-multifasta.c:270-277) and of genomic.txt.
 * The EST-sharded run (ef_multi.c) therefore gathers the RECORDS and the processed ESTs -- an eighth of the
 * bytes of the text -- and rank 0 prints the file from them with this routine. */
#include <stdlib.h>
#include <string.h>

#include "estfact.h"
#include "../../include/pintron_records.h"

/* appends the text of all records to `out`.  pests / pl: the processed-ests text of the same ESTs, in the same
 * order.  Returns 0, or -1 when the records are inconsistent with it. */
int ef_raw_text_from_records(const ef_seq* gen, const void* rec, size_t rl, const char* pests, size_t pl, ef_sink* out) {
  const char* G = gen->original_seq;
  const size_t gl = strlen(G);
  pfr_reader r; pfr_open(&r, rec, rl);
  pfr_est e; pfr_factorization f;
  const char* ep = pests; const char* const eend = pests + pl;
  ef_wbuf w; efw_open(&w, out);
  int rc;
  while ((rc = pfr_next_est(&r, &e)) == 1) {
    /* the next two lines of processed-ests.txt: ">header" and the sequence */
    if (ep >= eend) return -1;
    const char* h = ep; const char* nl = (const char*)memchr(ep, '\n', (size_t)(eend - ep));
    if (!nl) return -1;
    const size_t hl = (size_t)(nl - h); ep = nl + 1;
    const char* s = ep; nl = (const char*)memchr(ep, '\n', (size_t)(eend - ep));
    const size_t sl = nl ? (size_t)(nl - s) : (size_t)(eend - s);
    ep = nl ? nl + 1 : eend;
    while ((rc = pfr_next_factorization(&r, &f)) == 1) {
      efw_mem(&w, h, hl); efw_str(&w, "\n#polya="); efw_int(&w, f.polya); efw_str(&w, "\n#polyad="); efw_int(&w, f.polyad); efw_ch(&w, '\n');
      for (uint16_t k = 0; k < f.n_exons; ++k) {
        const pfr_exon x = pfr_exon_at(&f, k);
        efw_int(&w, x.est_start); efw_ch(&w, ' '); efw_int(&w, x.est_end); efw_ch(&w, ' ');
        efw_int(&w, x.gen_start); efw_ch(&w, ' '); efw_int(&w, x.gen_end); efw_ch(&w, ' ');
        /* "%.*s": at most end + 1 - start characters, fewer where the sequence ends first */
        if (x.est_start >= 1 && (size_t)x.est_start <= sl + 1 && x.est_end + 1 > x.est_start) {
          size_t n = (size_t)(x.est_end + 1 - x.est_start);
          if ((size_t)x.est_start - 1 + n > sl) n = sl - ((size_t)x.est_start - 1);
          efw_mem(&w, s + x.est_start - 1, n);
        }
        efw_ch(&w, ' ');
        if (x.gen_start >= 1 && (size_t)x.gen_start <= gl + 1 && x.gen_end + 1 > x.gen_start) {
          size_t n = (size_t)(x.gen_end + 1 - x.gen_start);
          if ((size_t)x.gen_start - 1 + n > gl) n = gl - ((size_t)x.gen_start - 1);
          efw_mem(&w, G + x.gen_start - 1, n);
        }
        efw_ch(&w, '\n');
      }
    }
    if (rc < 0) break;
  }
  efw_flush(&w);
  return rc < 0 ? -1 : 0;
}
