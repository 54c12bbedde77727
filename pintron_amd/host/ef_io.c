/* Multifasta input, sequence preparation and record output of est-fact.
 * Behaviour follows src/io-multifasta.c and src/main-est-fact.c of the reference (cited per
 * function); the implementation is our own. */
#define _GNU_SOURCE
#include <ctype.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#include "estfact.h"

/* ---- per-thread 32-byte cells (ef_list.h) ------------------------------------------------------- */
typedef struct cell_block { struct cell_block* next; ef_cell cells[2048]; } cell_block;
_Thread_local ef_cell* ef_cell_free_list;
_Thread_local long ef_cell_live;
static _Thread_local cell_block* cell_blocks;

void ef_cell_refill(void) {
  cell_block* b = (cell_block*)malloc(sizeof(cell_block));
  if (!b) { fprintf(stderr, "* FATAL out of memory\n"); abort(); }
  b->next = cell_blocks; cell_blocks = b;
  for (size_t i = 0; i < 2048; ++i) { b->cells[i].next = ef_cell_free_list; ef_cell_free_list = &b->cells[i]; }
}

typedef struct cell64_block { struct cell64_block* next; ef_cell64 cells[1024]; } cell64_block;
_Thread_local ef_cell64* ef_cell64_free_list;
static _Thread_local cell64_block* cell64_blocks;

void ef_cell64_refill(void) {
  cell64_block* b = (cell64_block*)malloc(sizeof(cell64_block));
  if (!b) { fprintf(stderr, "* FATAL out of memory\n"); abort(); }
  b->next = cell64_blocks; cell64_blocks = b;
  for (size_t i = 0; i < 1024; ++i) { b->cells[i].next = ef_cell64_free_list; ef_cell64_free_list = &b->cells[i]; }
}

void ef_cell_release_all(void) {
  if (ef_cell_live != 0) return;            /* something of this thread is still alive: keep the blocks */
  while (cell_blocks) { cell_block* nx = cell_blocks->next; free(cell_blocks); cell_blocks = nx; }
  while (cell64_blocks) { cell64_block* nx = cell64_blocks->next; free(cell64_blocks); cell64_blocks = nx; }
  ef_cell_free_list = NULL; ef_cell64_free_list = NULL;
}

/* ---- record arena ------------------------------------------------------------------------------- */
typedef struct rec_slab { struct rec_slab* next; void* base; size_t total, cap, used; } rec_slab;
struct ef_record_arena { rec_slab* slabs; pthread_mutex_t mu; };
static _Thread_local ef_record_arena* tl_arena;
static _Thread_local rec_slab* tl_slab;
enum { SLAB_BYTES = 16u << 20, HUGE_PAGE = 2u << 20 };

ef_record_arena* ef_record_arena_new(void) {
  ef_record_arena* a = (ef_record_arena*)calloc(1, sizeof *a);
  pthread_mutex_init(&a->mu, NULL);
  return a;
}
/* The slabs of a released arena wait for the next one (a process that brings one batch after the other: unmapping
 * 150 MB and faulting them in again cost 0.02 s per batch at either end); PINTRON_KEEP=0: not.  At most 1.5 GB wait. */
static struct { pthread_mutex_t mu; rec_slab* free; size_t n; } slab_cache = { PTHREAD_MUTEX_INITIALIZER, NULL, 0 };
static bool slab_keep(void) { static int k = -1; if (k < 0) { const char* e = getenv("PINTRON_KEEP"); k = !(e && e[0] == '0'); } return k != 0; }
void ef_record_arena_free(ef_record_arena* a) {
  if (!a) return;
  while (a->slabs) {
    rec_slab* sl = a->slabs;
    a->slabs = sl->next;
    bool kept = false;
    if (slab_keep() && sl->cap == SLAB_BYTES) {
      pthread_mutex_lock(&slab_cache.mu);
      if (slab_cache.n < 96) { sl->next = slab_cache.free; slab_cache.free = sl; ++slab_cache.n; kept = true; }
      pthread_mutex_unlock(&slab_cache.mu);
    }
    if (!kept) munmap(sl->base, sl->total);
  }
  pthread_mutex_destroy(&a->mu);
  free(a);
}
/* the arena's mappings, for a process that is about to end and gives its pages back from several threads */
size_t ef_record_arena_regions(ef_record_arena* a, void** base, size_t* len, size_t max) {
  size_t n = 0;
  if (!a) return 0;
  for (rec_slab* sl = a->slabs; sl && n < max; sl = sl->next) { base[n] = sl->base; len[n] = sl->total; ++n; }
  return n;
}
void ef_record_arena_enter(ef_record_arena* a) { tl_arena = a; tl_slab = NULL; }
void ef_record_arena_leave(void) { tl_arena = NULL; tl_slab = NULL; }

/* n bytes for a record of the calling thread: from its current slab of the arena it entered, else malloc */
static void* rec_alloc(size_t n) {
  if (!tl_arena) return malloc(n);
  n = (n + 15) & ~(size_t)15;
  if (!tl_slab || tl_slab->used + n > tl_slab->cap) {
    const size_t body = n + sizeof(rec_slab) + 64 > SLAB_BYTES ? n + sizeof(rec_slab) + 64 : SLAB_BYTES;
    const size_t total = body + HUGE_PAGE;                       /* room to start on a huge-page boundary */
    rec_slab* sl = NULL;
    if (body == SLAB_BYTES && slab_cache.n) {                    /* one that an earlier arena left */
      pthread_mutex_lock(&slab_cache.mu);
      if (slab_cache.free) { sl = slab_cache.free; slab_cache.free = sl->next; --slab_cache.n; }
      pthread_mutex_unlock(&slab_cache.mu);
    }
    if (!sl) {
      void* base = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (base == MAP_FAILED) return NULL;
      char* start = (char*)(((uintptr_t)base + HUGE_PAGE - 1) & ~(uintptr_t)(HUGE_PAGE - 1));
      if (getenv("PINTRON_ARENA_THP")) madvise(start, body & ~(size_t)(HUGE_PAGE - 1), MADV_HUGEPAGE);   /* experiment: direct compaction can stall */
      sl = (rec_slab*)start;
      sl->base = base; sl->total = total; sl->cap = body;
    }
    sl->used = (sizeof(rec_slab) + 15) & ~(size_t)15;
    pthread_mutex_lock(&tl_arena->mu);
    sl->next = tl_arena->slabs; tl_arena->slabs = sl;
    pthread_mutex_unlock(&tl_arena->mu);
    tl_slab = sl;
  }
  void* r = (char*)tl_slab + tl_slab->used;
  tl_slab->used += n;
  return r;
}

static char* dup_range(const char* s, size_t n) {
  char* r = (char*)rec_alloc(n + 1);
  memcpy(r, s, n);
  r[n] = '\0';
  return r;
}
static char* dup_str(const char* s) { return dup_range(s, strlen(s)); }

void ef_seq_free(ef_seq* s) {
  if (!s || s->in_arena) return;               /* an arena's records go with the arena */
  free(s->id); free(s->seq); free(s->original_seq); free(s->gb); free(s->chr);
  free(s->kmer_first); free(s->kmer_pos); free(s->bps_memo);
  free(s->other_first); free(s->other_pos);
  for (int k = 0; k < 4; ++k) free(s->score5_tab[k]);
  free(s->cls_start); free(s->cls_end);
  free(s);
}

/* 6-mer index of the (final) genomic working sequence: one counting sort over the sequence */
#define EF_KMER 6
static int kmer_base(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
unsigned ef_genomic_epoch = 1;
void ef_seq_index_kmers(ef_seq* gen) {
  ef_genomic_epoch_bump();                   /* a new genomic sequence is about to be used */
  const size_t n = strlen(gen->seq);
  free(gen->kmer_first); free(gen->kmer_pos); free(gen->bps_memo);
  gen->bps_memo = (unsigned char*)calloc(n + 2, 1);
  gen->kmer_first = (uint32_t*)calloc(4097, sizeof(uint32_t));
  gen->kmer_pos = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
  {                                          /* the bytes that are no upper-case ACGT, by value */
    free(gen->other_first); free(gen->other_pos);
    gen->other_first = (uint32_t*)calloc(257, sizeof(uint32_t));
    size_t n_other = 0;
    for (size_t i = 0; i < n; ++i) if (kmer_base(gen->seq[i]) < 0) { ++gen->other_first[(unsigned char)gen->seq[i] + 1]; ++n_other; }
    for (int c = 0; c < 256; ++c) gen->other_first[c + 1] += gen->other_first[c];
    gen->other_pos = (uint32_t*)malloc((n_other + 1) * sizeof(uint32_t));
    uint32_t fill[256];
    memcpy(fill, gen->other_first, sizeof fill);
    for (size_t i = 0; i < n; ++i) if (kmer_base(gen->seq[i]) < 0) gen->other_pos[fill[(unsigned char)gen->seq[i]]++] = (uint32_t)i;
  }
  if (n < EF_KMER) return;
  const size_t nk = n - EF_KMER + 1;
  int* codes = (int*)malloc(nk * sizeof(int));
  for (size_t i = 0; i < nk; ++i) {
    int code = 0;
    for (int k = 0; k < EF_KMER && code >= 0; ++k) { const int b = kmer_base(gen->seq[i + k]); code = b < 0 ? -1 : ((code << 2) | b); }
    codes[i] = code;
    if (code >= 0) ++gen->kmer_first[code + 1];
  }
  for (int c = 0; c < 4096; ++c) gen->kmer_first[c + 1] += gen->kmer_first[c];
  uint32_t* fill = (uint32_t*)malloc(4096 * sizeof(uint32_t));
  memcpy(fill, gen->kmer_first, 4096 * sizeof(uint32_t));
  for (size_t i = 0; i < nk; ++i) if (codes[i] >= 0) gen->kmer_pos[fill[codes[i]]++] = (uint32_t)i;
  free(fill); free(codes);
}

static ef_seq* seq_new(void) {
  ef_seq* s = (ef_seq*)rec_alloc(sizeof(ef_seq));
  memset(s, 0, sizeof *s);
  s->strand = 1;
  s->in_arena = tl_arena != NULL;
  return s;
}

/* One logical line: bytes up to '\n', then trailing bytes whose SIGNED value is below ' ' are
 * dropped (src/util.c:166-173 strips with a plain `char` comparison, so bytes >= 0x80 go too). */
typedef struct { const char* p; size_t len; } line_t;

static bool next_line(const char* buf, size_t size, size_t* pos, line_t* ln) {
  if (*pos >= size) return false;
  const char* start = buf + *pos;
  const char* nl = (const char*)memchr(start, '\n', size - *pos);
  size_t raw = nl ? (size_t)(nl - start) + 1 : size - *pos;
  *pos += raw;
  while (raw > 0 && (signed char)start[raw - 1] < ' ') --raw;
  ln->p = start; ln->len = raw;
  return true;
}

/* read_multifasta (src/io-multifasta.c:133-164) with getData (:94-130): a record starts at a line
 * beginning with '>', its sequence is the concatenation of the following non-empty lines up to
 * the next '>' line, the literal line "#\#" or the end of the file. */
/* the records of buf[0, sz), which starts at a record (or before the first one) */
typedef struct { const char* buf; size_t sz; ef_seq** v; size_t n; ef_record_arena* arena; } parse_range;
static void* parse_records(void* arg) {
  parse_range* r = (parse_range*)arg;
  const char* buf = r->buf; const size_t sz = r->sz;
  size_t pos = 0, cap = 16, n = 0;
  ef_seq** v = (ef_seq**)malloc(cap * sizeof(ef_seq*));
  ef_record_arena* before = tl_arena; rec_slab* slab_before = tl_slab;
  if (r->arena) ef_record_arena_enter(r->arena);
  size_t scap = 4096;
  char* data = (char*)malloc(scap);             /* the lines of one record are gathered here, then copied at their size */
  line_t ln;
  bool have = next_line(buf, sz, &pos, &ln);
  while (have) {
    if (ln.len > 0 && ln.p[0] == '>') {
      ef_seq* s = seq_new();
      s->id = dup_range(ln.p + 1, ln.len - 1);
      size_t slen = 0;
      while ((have = next_line(buf, sz, &pos, &ln))) {
        if (ln.len > 0 && ln.p[0] == '>') break;
        if (ln.len == 3 && memcmp(ln.p, "#\\#", 3) == 0) break;
        if (ln.len == 0) continue;
        /* an embedded NUL would end the reference's C string copy of the line */
        size_t use = strnlen(ln.p, ln.len);
        if (slen + use + 1 > scap) { while (slen + use + 1 > scap) scap *= 2; data = (char*)realloc(data, scap); }
        memcpy(data + slen, ln.p, use);
        slen += use;
      }
      s->seq = dup_range(data, slen);
      s->original_seq = dup_range(data, slen);
      if (n == cap) { cap *= 2; v = (ef_seq**)realloc(v, cap * sizeof(ef_seq*)); }
      v[n++] = s;
    } else {
      have = next_line(buf, sz, &pos, &ln);
    }
  }
  free(data);
  tl_arena = before; tl_slab = slab_before;
  r->v = v; r->n = n;
  return NULL;
}

/* A record starts at every line that begins with '>', whatever came before it, so a large file is
 * cut at such lines into a few ranges that are parsed side by side (a C5-sized ests.txt holds two
 * million records: one thread spends most of its time in malloc) and joined in file order. */
int ef_parse_threads = 8;                    /* the scheduler sets it to the process's share of the cores */
enum { MAX_PARTS = 32 };

/* the cut of buf[0, sz) at or after `at`: behind the first newline at or after `at` that is followed by '>' */
static size_t cut_in_buffer(const char* buf, size_t sz, size_t at) {
  const char* q = buf + at;
  while ((q = (const char*)memchr(q, '\n', (size_t)(buf + sz - q))) != NULL && q + 1 < buf + sz && q[1] != '>') ++q;
  return (!q || q + 1 >= buf + sz) ? sz : (size_t)(q + 1 - buf);
}

static long parse_buffer(const char* buf, size_t sz, ef_record_arena* arena, ef_seq*** out) {
  int want = (sz >= (8u << 20) || getenv("PINTRON_PARSE_SPLIT")) ? ef_parse_threads : 1;   /* the variable: tests */
  if (want > MAX_PARTS) want = MAX_PARTS;
  if (want < 1) want = 1;
  size_t cut[MAX_PARTS + 1];
  int parts = 0;
  cut[0] = 0;
  for (int t = 1; t < want; ++t) {
    const size_t at = sz * (size_t)t / (size_t)want;
    if (at <= cut[parts]) continue;
    const size_t c = cut_in_buffer(buf, sz, at);
    if (c >= sz) break;
    cut[++parts] = c;
  }
  cut[++parts] = sz;
  parse_range rg[MAX_PARTS]; pthread_t th[MAX_PARTS]; bool started[MAX_PARTS];
  for (int t = 0; t < parts; ++t) {
    rg[t].buf = buf + cut[t]; rg[t].sz = cut[t + 1] - cut[t]; rg[t].v = NULL; rg[t].n = 0; rg[t].arena = arena;
    started[t] = parts > 1 && pthread_create(&th[t], NULL, parse_records, &rg[t]) == 0;
    if (!started[t]) parse_records(&rg[t]);
  }
  size_t n = 0;
  for (int t = 0; t < parts; ++t) { if (started[t]) pthread_join(th[t], NULL); n += rg[t].n; }
  ef_seq** v = (ef_seq**)malloc((n + 1) * sizeof(ef_seq*));
  size_t at = 0;
  for (int t = 0; t < parts; ++t) { memcpy(v + at, rg[t].v, rg[t].n * sizeof(ef_seq*)); at += rg[t].n; free(rg[t].v); }
  *out = v;
  return (long)n;
}

/* bytes [a, b) of the file: mapped (no copy out of the page cache, the pages entered in one go), read into a
 * block where mapping fails.  The parser takes sizes, never a terminator. */
typedef struct { const char* p; size_t len; void* map; size_t map_len; char* heap; } file_bytes;
static int file_bytes_open(const char* path, long a, long b, file_bytes* fb) {
  memset(fb, 0, sizeof *fb);
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return -1;
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return -1; }
  if (b < 0 || b > (long)sb.st_size) b = (long)sb.st_size;
  if (a > b) a = b;
  fb->len = (size_t)(b - a);
  if (fb->len == 0) { close(fd); fb->p = ""; return 0; }
  const long page = sysconf(_SC_PAGESIZE);
  const long a0 = a & ~(page - 1);
  void* m = mmap(NULL, (size_t)(b - a0), PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, a0);
  if (m != MAP_FAILED) { fb->map = m; fb->map_len = (size_t)(b - a0); fb->p = (const char*)m + (a - a0); close(fd); return 0; }
  fb->heap = (char*)malloc(fb->len + 1);
  const ssize_t got = fb->heap ? pread(fd, fb->heap, fb->len, a) : -1;
  close(fd);
  if (got != (ssize_t)fb->len) { free(fb->heap); return -1; }
  fb->p = fb->heap;
  return 0;
}
static void file_bytes_close(file_bytes* fb) { if (fb->map) munmap(fb->map, fb->map_len); free(fb->heap); }

long ef_read_multifasta(const char* path, ef_seq*** out) { return ef_read_multifasta_arena(path, 0, 1, NULL, out); }
long ef_read_multifasta_part(const char* path, int part, int parts, ef_seq*** out) { return ef_read_multifasta_arena(path, part, parts, NULL, out); }

/* Part `part` of `parts` of the file (0 of 1: all of it), for the EST-sharded runs: the records that begin in
 * [cut(part), cut(part + 1)), where cut(t) is the first record start behind a newline at or after byte
 * size * t / parts (cut(0) = 0, cut(parts) = size) -- the same rule on every rank, so the parts are disjoint,
 * in file order, and together the whole file; and a rank reads and parses only its own bytes (every rank
 * parsing all two million reads of a C5-sized file cost each of them 1.5 s for a step of 0.2 s).  The records
 * go to `arena` when there is one, else to malloc. */
static long file_cut(FILE* f, long sz, long at) {
  if (at <= 0) return 0;
  if (at >= sz) return sz;
  enum { WIN = 1 << 20 };
  char* w = (char*)malloc(WIN + 1);
  long found = sz;
  for (long pos = at; pos < sz && found == sz;) {
    fseek(f, pos, SEEK_SET);
    const size_t got = fread(w, 1, WIN + 1, f);                  /* one byte of overlap: the '>' behind a newline at the window's end */
    if (got == 0) break;
    for (size_t k = 0; k + 1 < got; ++k) if (w[k] == '\n' && w[k + 1] == '>') { found = pos + (long)k + 1; break; }
    if (got < (size_t)WIN + 1) break;
    pos += WIN;
  }
  free(w);
  return found;
}

long ef_read_multifasta_arena(const char* path, int part, int parts, ef_record_arena* arena, ef_seq*** out) {
  long a = 0, b = -1;
  if (parts > 1) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    a = file_cut(f, sz, (long)((unsigned long long)sz * (unsigned)part / (unsigned)parts));
    b = part + 1 >= parts ? sz : file_cut(f, sz, (long)((unsigned long long)sz * (unsigned)(part + 1) / (unsigned)parts));
    fclose(f);
  }
  file_bytes fb;
  if (file_bytes_open(path, a, b, &fb) != 0) return -1;
  const long n = parse_buffer(fb.p, fb.len, arena, out);
  file_bytes_close(&fb);
  return n;
}

/* parse_genomic_header (src/io-multifasta.c:307-423): ">chr:start:end:+-1", else defaults */
void ef_parse_genomic_header(ef_seq* gen) {
  char* h = strdup(gen->id);
  char* fields[5] = {0};
  int nf = 0;
  char* cur = h;
  while (cur && nf < 5) { fields[nf++] = strsep(&cur, ":"); }
  bool ok = (nf == 4 && cur == NULL);
  if (ok) {
    const int a = atoi(fields[1]), b = atoi(fields[2]), st = atoi(fields[3]);
    ok = a >= 1 && b >= 1 && (st == -1 || st == 1);
    if (ok) {
      gen->chr = strdup(fields[0]);
      gen->abs_start = a; gen->abs_end = b; gen->strand = st;
      snprintf(gen->strand_as_read, sizeof gen->strand_as_read, "%s", fields[3]);
    }
  }
  if (!ok) {
    fprintf(stderr, "* ERROR The header of the genomic file is not in the correct standard! "
                    "This may lead to prediction errors.\n");
    gen->chr = strdup("unknown");
    gen->abs_start = 1; gen->abs_end = (int)strlen(gen->seq); gen->strand = 1;
    strcpy(gen->strand_as_read, "+1");
  }
  free(h);
}

/* Ntails_removal (src/io-multifasta.c:830-868): upper-case N runs at both ends of the working
 * sequence; the original sequence keeps them and pref_N_length shifts the output coordinates. */
int ef_ntails_removal(ef_seq* gen) {
  char* s = gen->seq;
  size_t len = strlen(s), pref = 0, suff = 0;
  while (s[pref] == 'N') ++pref;
  if (pref) memmove(s, s + pref, len - pref + 1);
  gen->pref_N_length = (int)pref;
  len -= pref;
  while (suff < len && s[len - 1 - suff] == 'N') ++suff;
  if (suff == len) return -1;
  s[len - suff] = '\0';
  gen->suff_N_length = (int)suff;
  return 0;
}

/* set_EST_GB_identification (src/io-multifasta.c:279-304) */
void ef_set_gb_identification(ef_seq* est) {
  const char* p = strstr(est->id, "/gb=");
  if (!p) p = strstr(est->id, "/GB=");
  if (!p) return;
  p += 4;
  size_t len = 0;
  while (p[len] != ' ' && p[len] != '/' && p[len] != '\0') ++len;
  est->gb = dup_range(p, len);
}

static char complement(char c) {           /* get_Complement (src/io-multifasta.c:40-92) */
  static const char from[] = "ATCGRYMKBVDHatcgrymkbvdh";
  static const char to[]   = "TAGCYRKMVBHDtagcyrkmvbhd";
  const char* q = c ? strchr(from, c) : NULL;
  return q ? to[q - from] : c;
}

/* reverse_and_complement (src/io-multifasta.c:506-522).  Both sequences are rewritten from the
 * WORKING sequence, so a sibling built after polyA/T masking carries the mask characters in its
 * output sequence too. */
void ef_reverse_and_complement(ef_seq* est) {
  const size_t len = strlen(est->seq);
  if (len == 0) return;
  size_t left = 0, right = len - 1;
  while (left <= right) {
    const char nr = complement(est->seq[left]), nl = complement(est->seq[right]);
    est->seq[right] = nr; est->seq[left] = nl;
    est->original_seq[right] = nr; est->original_seq[left] = nl;
    ++left;
    if (right == 0) break;
    --right;
  }
}

/* set_EST_Strand_and_RC (src/io-multifasta.c:425-504) */
void ef_set_strand_and_rc(ef_seq* est) {
  est->strand_as_read[0] = '\0';
  const char* gb = est->gb;
  const bool refseq = gb && gb[0] == 'N' && gb[1] != '\0' && gb[2] == '_' && (gb[1] == 'M' || gb[1] == 'R');
  if (refseq) {
    strcpy(est->strand_as_read, "1");
    est->strand = 1;
    est->fixed_strand = true;
  } else {
    const char* p = strstr(est->id, "/clone_end=");
    if (!p) p = strstr(est->id, "/CLONE_END=");
    if (p) {
      p += 11;
      int i = 0;
      while (i < 10 && *p != '\0' && *p != '\'') est->strand_as_read[i++] = *p++;
      est->strand_as_read[i] = '\0';
      bool valid = false;
      if (!strcmp(est->strand_as_read, "3")) { est->strand = 1; valid = true; }
      else if (!strcmp(est->strand_as_read, "5")) { est->strand = -1; valid = true; }
      else est->strand = 1;
      if (valid) {
        p = strstr(est->id, "/fixed_strand=");
        if (!p) p = strstr(est->id, "/FIXED_STRAND=");
        if (p) est->fixed_strand = (p[14] == '1');
      }
    } else {
      est->strand = 1;
    }
  }
  if (est->strand == -1) ef_reverse_and_complement(est);
}

/* one end of polyAT_substitution (src/io-multifasta.c:662-828); at(i) walks inwards from the end */
#define POLYA_MIN_LEN 14
#define POLYA_MIN_FRACTION 0.72
static int mask_end(char* s, size_t len, bool from_back, char* which) {
#define AT(i) (*(from_back ? &s[len - 1 - (i)] : &s[(i)]))
  size_t count_A = 0, count_T = 0, last_A = 0, last_T = 0, last_A_count = 0, last_T_count = 0;
  size_t i;
  for (i = 0; i < POLYA_MIN_LEN && i < len; ++i) {
    if (AT(i) == 'A') { ++count_A; last_A = i; last_A_count = count_A; }
    if (AT(i) == 'T') { ++count_T; last_T = i; last_T_count = count_T; }
  }
  size_t run_A = count_A, run_T = count_T;
  while (i < len && (run_A >= (POLYA_MIN_FRACTION * POLYA_MIN_LEN) || run_T >= (POLYA_MIN_FRACTION * POLYA_MIN_LEN))) {
    if (AT(i - POLYA_MIN_LEN) == 'A') --run_A;
    if (AT(i - POLYA_MIN_LEN) == 'T') --run_T;
    if (AT(i) == 'A') { ++count_A; ++run_A; last_A = i; last_A_count = count_A; }
    if (AT(i) == 'T') { ++count_T; ++run_T; last_T = i; last_T_count = count_T; }
    ++i;
  }
  if (last_A < POLYA_MIN_LEN - 1) last_A = POLYA_MIN_LEN - 1;
  if (last_T < POLYA_MIN_LEN - 1) last_T = POLYA_MIN_LEN - 1;
  if (last_A_count >= (POLYA_MIN_FRACTION * (last_A + 1)) || last_T_count >= (POLYA_MIN_FRACTION * (last_T + 1))) {
    const char c = (((double)last_A_count) / (last_A + 1)) >= (((double)last_T_count) / (last_T + 1)) ? 'A' : 'T';
    const size_t mlen = c == 'A' ? last_A + 1 : last_T + 1;
    for (i = 0; i < mlen; ++i) AT(i) = c == 'A' ? EF_POLYA_CHR : EF_POLYT_CHR;
    *which = c;
    return (int)mlen;
  }
  return -1;
#undef AT
}

void ef_polyAT_substitution(ef_seq* est) {
  const size_t len = strlen(est->seq);
  est->pref_polyA_length = est->suff_polyA_length = -1;
  est->pref_polyT_length = est->suff_polyT_length = -1;
  if (len < POLYA_MIN_LEN) return;
  char c = 0;
  int m = mask_end(est->seq, len, false, &c);
  if (m >= 0) { if (c == 'A') est->pref_polyA_length = m; else est->pref_polyT_length = m; }
  m = mask_end(est->seq, len, true, &c);
  if (m >= 0) { if (c == 'A') est->suff_polyA_length = m; else est->suff_polyT_length = m; }
}

/* copy_and_reverse (src/main-est-fact.c:67-87) */
ef_seq* ef_copy_and_reverse(const ef_seq* est) {
  ef_seq* r = seq_new();
  r->seq = dup_str(est->seq);
  r->original_seq = dup_str(est->original_seq);
  ef_reverse_and_complement(r);
  r->id = dup_str(est->id);
  r->gb = est->gb ? dup_str(est->gb) : NULL;
  r->chr = est->chr ? dup_str(est->chr) : NULL;
  memcpy(r->strand_as_read, est->strand_as_read, sizeof r->strand_as_read);
  r->strand = -est->strand;
  r->fixed_strand = est->fixed_strand;
  r->pref_polyA_length = est->suff_polyT_length;
  r->suff_polyA_length = est->pref_polyT_length;
  r->pref_polyT_length = est->suff_polyA_length;
  r->suff_polyT_length = est->pref_polyA_length;
  return r;
}

void ef_write_single_est_info(ef_sink* f, const ef_seq* s) {
  ef_wbuf w; efw_open(&w, f);
  efw_ch(&w, '>'); efw_str(&w, s->id); efw_ch(&w, '\n'); efw_str(&w, s->original_seq); efw_ch(&w, '\n');
  efw_flush(&w);
}
