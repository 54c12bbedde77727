/* Packed factorization records of est-fact (MI355X build): everything the downstream stages parse
 * out of raw-multifasta-out.txt, without the text.
 *
 * Replaces, on the consumer side, the text parser of the reference's min-factorization /
 * intron-agreement input (src/io-factorizations.c:128-231: "%d %d %d %d" exon lines, "#polya=",
 * "#polyad=", grouping by header); SURVEY.md section 8f.1.  Producer:
 * pintron_amd/host/ef_estfact.c (ef_write_factorization_records); est-fact writes the blob when
 * PINTRON_RECORDS_FILE=<path> is set, sessions hand it out as output 6, and `bench.py --gpus N`
 * gathers it to rank 0 over RCCL.
 *
 * Layout (little-endian, unaligned), one group per ALIGNED EST -- per entry of processed-ests.txt, in the
 * same (input) order; n_factorizations is 0 for an EST whose factorizations --retain-externals=false all
 * dropped (src/io-multifasta.c:204,217-222: it has no line in raw-multifasta-out.txt) --:
 *   u32 est_index            position of the EST in ests.txt (0-based)
 *   u32 n_factorizations
 *   per factorization:  u8 polya, u8 polyad, u16 n_exons,
 *                       n_exons x { i32 EST_start, EST_end, GEN_start, GEN_end }
 * Coordinates are exactly the numbers of the text file: 1-based, inclusive, genomic coordinates
 * relative to the sequence as it stands in genomic.txt.
 *
 * Header-only reader, plain C99, no allocation.  Every function returns 1 on success, 0 at the end
 * of the data, -1 when the blob is truncated or inconsistent. */
#ifndef PINTRON_RECORDS_H
#define PINTRON_RECORDS_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct {
  const unsigned char* p;      /* read position */
  const unsigned char* end;
  uint32_t facts_left;         /* factorizations of the current EST not yet read */
} pfr_reader;

typedef struct { uint32_t est_index, n_factorizations; } pfr_est;
typedef struct { int32_t est_start, est_end, gen_start, gen_end; } pfr_exon;
typedef struct {
  uint8_t polya, polyad;
  uint16_t n_exons;
  const unsigned char* exons;  /* n_exons x 16 bytes, unaligned: read with pfr_exon_at */
} pfr_factorization;

static inline void pfr_open(pfr_reader* r, const void* data, size_t len) {
  r->p = (const unsigned char*)data; r->end = r->p + len; r->facts_left = 0;
}

/* next EST group; the factorizations of the previous one must have been read (or skipped with
 * pfr_next_factorization until it returns 0) */
static inline int pfr_next_est(pfr_reader* r, pfr_est* e) {
  if (r->facts_left != 0) return -1;
  if (r->p == r->end) return 0;
  if ((size_t)(r->end - r->p) < 8) return -1;
  memcpy(&e->est_index, r->p, 4); memcpy(&e->n_factorizations, r->p + 4, 4);
  r->p += 8;
  r->facts_left = e->n_factorizations;
  return 1;
}

/* next factorization of the current EST; 0 when the EST has no more */
static inline int pfr_next_factorization(pfr_reader* r, pfr_factorization* f) {
  if (r->facts_left == 0) return 0;
  if ((size_t)(r->end - r->p) < 4) return -1;
  f->polya = r->p[0]; f->polyad = r->p[1];
  memcpy(&f->n_exons, r->p + 2, 2);
  r->p += 4;
  if ((size_t)(r->end - r->p) < (size_t)f->n_exons * 16u) return -1;
  f->exons = r->p;
  r->p += (size_t)f->n_exons * 16u;
  --r->facts_left;
  return 1;
}

static inline pfr_exon pfr_exon_at(const pfr_factorization* f, uint16_t k) {
  pfr_exon x;
  memcpy(&x, f->exons + (size_t)k * 16u, 16);
  return x;
}

#endif
