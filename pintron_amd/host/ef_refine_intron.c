/* Intron boundary refinement on the host: window extraction, canonical-site search on the gapped
 * alignment, the four shift heuristics and the Burset fallback.
 * Behaviour follows src/refine-intron.c:47-344 and :892-1983 of the reference (cited per function).
 * The 3-state gap alignment itself and every edit distance are backend (GPU) calls. */
#include <stdlib.h>
#include <string.h>

#include "estfact.h"

typedef struct {
  char* est_row; char* gen_row;     /* the zero-padded alignment rows (one block, est_row first) */
  int dim, factor_cut, intron_start, intron_end, intron_start_on_align, intron_end_on_align;
  int new_acceptor_factor_left, new_donor_right_on_gen, new_acceptor_left_on_gen;
} gap_aln;


/* ---- Burset frequencies ----------------------------------------------------------------------- */
/* getBursetFrequency (src/refine-intron.c:376-556) as data: index = donor[0],donor[1],acceptor[0],
 * acceptor[1] at 2 bits each (A=0 C=1 G=2 T=3) */
static const unsigned char burset_tab[256] = {
    0,   0,   1,   1,   0,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,
    0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   1,   5,   0,   0,   0,   0,   2,   0,   1,   0,   0,   0,   0,   2,   0,
    1,   8,   7,   2,   0,   0,   0,   0,   0,   1,   0,   1,   0,   0,   0,   0,
    0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   1,
    0,   0,   2,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   0,   1,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   2,   0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
    0,   0,   8,   0,   0,   0,   0,   0,   0,   0,   0,   1,   0,   1,   1,   0,
    0,   0, 126,   0,   0,   0,   0,   0,   0,   0,   1,   0,   1,   0,   0,   0,
    0,   1,  11,   0,   1,   0,   0,   0,   2,   0,   0,   0,   0,   2,   0,   0,
    0,   4, 200,   2,   9,   0,   4,   3,   0,   1,  10,   1,   7,   2,   8,   2,
    0,   0,   6,   0,   0,   0,   1,   0,   0,   0,   0,   0,   0,   1,   0,   0,
    0,   0,   1,   0,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,
    0,   1,   7,   0,   0,   0,   0,   0,   0,   0,   2,   0,   0,   0,   0,   0,
    0,   0,   5,   1,   0,   0,   0,   0,   0,   0,   1,   0,   0,   0,   0,   0,
};

static int code(char c) {
  switch (c) {
    case 'A': case 'a': return 0; case 'C': case 'c': return 1;
    case 'G': case 'g': return 2; case 'T': case 't': return 3;
    default: return -1;
  }
}

int ef_burset_frequency(const char* donor, const char* acceptor) {
  if (strlen(donor) != 2 || strlen(acceptor) != 2) return 0;
  const int a = code(donor[0]), b = code(donor[1]), c = code(acceptor[0]), d = code(acceptor[1]);
  if ((a | b | c | d) < 0) return 0;
  return burset_tab[(a << 6) | (b << 4) | (c << 2) | d];
}

int ef_burset_adaptor(const char* t, size_t cut1, size_t cut2) {        /* :362-374 */
  if (cut2 < 2) return 0;
  char d[3] = { t[cut1], 0, 0 }, a[3] = { t[cut2 - 2], t[cut2 - 1], 0 };
  if (d[0]) d[1] = t[cut1 + 1];
  return ef_burset_frequency(d, a);
}

int ef_check_burset_patterns(const char* gen, int donor_left, int acceptor_right) {   /* :346-360 */
  /* real_substring(donor_left + 1, 2, gen) and real_substring(acceptor_right - 2, 2, gen) (clamped
   * at the start of the sequence, cut at its terminator) without the two allocations */
  char d[3] = { 0, 0, 0 }, a[3] = { 0, 0, 0 };
  int di = donor_left + 1, dl = 2, ai = acceptor_right - 2, al = 2;
  if (di < 0) { dl += di; di = 0; }
  if (ai < 0) { al += ai; ai = 0; }
  for (int k = 0; k < dl && gen[di + k] != '\0'; ++k) d[k] = gen[di + k];
  for (int k = 0; k < al && gen[ai + k] != '\0'; ++k) a[k] = gen[ai + k];
  return ef_burset_frequency(d, a);
}

/* ---- searches on the gapped alignment ---------------------------------------------------------- */
/* Find_AG_after_on_the_right (:892-940) */
static void find_AG_after_right(const gap_aln* al, int init, int* cut_on_align, int* gen_cut, int* est_cut) {
  *cut_on_align = -1; *gen_cut = -1; *est_cut = -1;
  size_t index = (size_t)(init - 2);
  const size_t glen = strlen(al->gen_row);
  bool stop = false;
  while (!stop && index < glen - 1) {
    while (al->gen_row[index] == '-') ++index;
    char pt[3];
    pt[0] = al->gen_row[index];
    ++index;
    while (al->gen_row[index] == '-') ++index;
    pt[1] = al->gen_row[index];
    pt[2] = '\0';
    stop = strcmp(pt, "AG") == 0;
  }
  if (!stop) return;
  int cg = 0, ce = 0;
  *cut_on_align = (int)index + 1;
  for (size_t i = (size_t)(al->intron_end_on_align + 1); i <= index; ++i) {
    if (al->gen_row[i] != '-') ++cg;
    if (al->est_row[i] != '-') ++ce;
  }
  *gen_cut = cg; *est_cut = ce;
}

/* Find_ACCEPTOR_before_on_the_left (:942-990) */
static void find_before_left(const gap_aln* al, int init, int* cut_on_align, int* gen_cut, int* est_cut, const char* pat) {
  *cut_on_align = -1; *gen_cut = -1; *est_cut = -1;
  int index = init + 2;
  bool stop = false;
  while (!stop && index > 0) {
    while (al->gen_row[index] == '-') --index;
    char pt[3];
    pt[1] = al->gen_row[index];
    --index;
    while (index >= 0 && al->gen_row[index] == '-') --index;
    pt[0] = index < 0 ? '\0' : al->gen_row[index];
    pt[2] = '\0';
    if (strcmp(pt, pat) == 0) stop = true;
  }
  if (!stop) return;
  int cg = 0, ce = 0;
  *cut_on_align = index - 1;
  for (int i = al->intron_start_on_align - 1; i >= index; --i) {
    if (al->gen_row[i] != '-') ++cg;
    if (al->est_row[i] != '-') ++ce;
  }
  *gen_cut = cg; *est_cut = ce;
}

/* Find_ACCEPTOR_after_on_the_left (:1852-1874) */
static void find_after_left(const gap_aln* al, int init, int* substr_dim, const char* pat) {
  *substr_dim = -1;
  int index = init;
  bool stop = false;
  while (!stop && index < al->intron_end_on_align) {
    char pt[3];
    pt[0] = al->gen_row[index];
    ++index;
    pt[1] = al->gen_row[index];
    pt[2] = '\0';
    if (strcmp(pt, pat) == 0) stop = true;
  }
  if (!stop) return;
  *substr_dim = index - al->intron_start_on_align - 1;
}

/* Find_AG_before_on_the_right (:1950-1972) */
static void find_AG_before_right(const gap_aln* al, int init, int* substr_dim) {
  *substr_dim = -1;
  int index = init;
  bool stop = false;
  while (!stop && index > al->intron_start_on_align) {
    char pt[3];
    pt[1] = al->gen_row[index];
    --index;
    pt[0] = al->gen_row[index];
    pt[2] = '\0';
    if (strcmp(pt, "AG") == 0) stop = true;
  }
  if (!stop) return;
  *substr_dim = al->intron_end_on_align - index - 1;
}

/* the dozen short strings of one Shift_* call are carved from a block on the (fibre) stack */
typedef struct { char buf[1024]; size_t used; } strpool;
static char* sp_get(strpool* sp, size_t n) {
  if (sp->used + n <= sizeof sp->buf) { char* r = sp->buf + sp->used; sp->used += n; return r; }
  return (char*)malloc(n);
}
static void sp_put(strpool* sp, char* p) {
  if (p && !(p >= sp->buf && p < sp->buf + sizeof sp->buf)) free(p);
}
/* real_substring (src/util.c:138-158) into the pool */
static char* sp_substring(strpool* sp, int index, int length, const char* s) {
  if (index < 0) { length += index; index = 0; }
  if (length < 0) length = 0;
  char* r = sp_get(sp, (size_t)length + 1);
  int k = 0;
  while (k < length && s[index + k] != '\0') { r[k] = s[index + k]; ++k; }
  r[k] = '\0';
  return r;
}

/* Get_genomic/est_substring_from_alignment (:1878-1948): ungapped row segment + mismatch count */
static char* row_substring(strpool* sp, const gap_aln* al, bool genomic, int init, int length, int* error) {
  const int glen = (int)strlen(al->gen_row);
  if (init < 0 || init >= glen) return NULL;
  const int rlen = (int)strlen(genomic ? al->gen_row : al->est_row);
  const int actual = (rlen - init < length) ? rlen - init : length;
  char* out = sp_get(sp, (size_t)(actual > 0 ? actual : 0) + 1);
  int k = 0, herr = 0;
  const char* row = genomic ? al->gen_row : al->est_row;
  for (int index = init; index < init + actual; ++index) {
    if (row[index] != '-') out[k++] = row[index];
    if (al->gen_row[index] != al->est_row[index]) ++herr;
  }
  out[k] = '\0';
  *error = herr;
  return out;
}

static char* concat(strpool* sp, const char* a, const char* b) {
  char* r = sp_get(sp, strlen(a) + strlen(b) + 1);
  strcpy(r, a); strcat(r, b);
  return r;
}

/* Common body of the four Shift_* routines (:992-1850).
 *   r2l      true: search AG to the right of the intron and the donor pattern inside it (3'->5')
 *   variant1 true: "_1" decision rule (GT), false: "_2" rule (GC)                                   */
#define CYCLES 2
static bool shift_generic(const char* est, const char* gen, const gap_aln* al, bool r2l, bool variant1,
                          const char* pat, int* out_donor_right, int* out_acc_left, int* out_factor_left,
                          ef_backend* be) {
  int init_right = r2l ? al->intron_end_on_align + 1 : al->intron_end_on_align;
  int init_left = r2l ? al->intron_start_on_align : al->intron_start_on_align - 1;
  int cut_on_align = 0;
  int gen_cut[CYCLES], est_cut[CYCLES], sub_dim[CYCLES];
  char *cut_factor[CYCLES], *match_str[CYCLES], *prev_match[CYCLES], *ext_cut[CYCLES], *ext_match[CYCLES];
  int ext_error = -1;
  char *ext_est = NULL, *ext_gen = NULL;
  strpool pool; pool.used = 0;
  strpool* sp = &pool;
  if (r2l) {
    int l_substr = 8, start = al->intron_start_on_align - l_substr;
    if (start < 0) { l_substr = l_substr - start; start = 0; }
    ext_est = row_substring(sp, al, false, start, l_substr, &ext_error);
    ext_gen = row_substring(sp, al, true, start, l_substr, &ext_error);
  } else {
    ext_est = row_substring(sp, al, false, al->intron_end_on_align + 1, 8, &ext_error);
    ext_gen = row_substring(sp, al, true, al->intron_end_on_align + 1, 8, &ext_error);
  }
  for (int i = 0; i < CYCLES; ++i) {
    if (r2l) find_AG_after_right(al, init_right, &cut_on_align, &gen_cut[i], &est_cut[i]);
    else find_before_left(al, init_left, &cut_on_align, &gen_cut[i], &est_cut[i], pat);
    prev_match[i] = NULL; cut_factor[i] = NULL; ext_cut[i] = NULL;
    if (est_cut[i] > -1) {
      if (r2l) {
        prev_match[i] = sp_substring(sp, al->new_acceptor_left_on_gen, gen_cut[i], gen);
        cut_factor[i] = sp_substring(sp, al->new_acceptor_factor_left, est_cut[i], est);
        init_right = cut_on_align + 1;
      } else {
        prev_match[i] = sp_substring(sp, al->new_donor_right_on_gen - gen_cut[i] + 1, gen_cut[i], gen);
        cut_factor[i] = sp_substring(sp, al->new_acceptor_factor_left - est_cut[i], est_cut[i], est);
        init_left = cut_on_align - 1;
      }
      if (ext_error > 0 && ext_est != NULL)
        ext_cut[i] = r2l ? concat(sp, ext_est, cut_factor[i]) : concat(sp, cut_factor[i], ext_est);
    }
    if (r2l) find_after_left(al, init_left, &sub_dim[i], pat);
    else find_AG_before_right(al, init_right, &sub_dim[i]);
    match_str[i] = NULL; ext_match[i] = NULL;
    if (sub_dim[i] > -1) {
      if (r2l) {
        match_str[i] = sp_substring(sp, al->new_donor_right_on_gen + 1, sub_dim[i], gen);
        init_left = al->intron_start_on_align + sub_dim[i] + 1;
      } else {
        match_str[i] = sp_substring(sp, al->new_acceptor_left_on_gen - sub_dim[i], sub_dim[i], gen);
        init_right = al->intron_end_on_align - sub_dim[i] - 1;
      }
      if (cut_factor[i] != NULL && ext_error > 0 && ext_gen != NULL)
        ext_match[i] = r2l ? concat(sp, ext_gen, match_str[i]) : concat(sp, match_str[i], ext_gen);
    }
  }
  sp_put(sp, ext_est); sp_put(sp, ext_gen);

  /* Every edit distance the decision loops below can ask for is a function of strings that are
   * fixed by now (<= 2 + 4 pairs of at most a few dozen characters): they are requested together
   * and the loops read the answers, instead of suspending the EST up to six times. */
  uint32_t ed_prev[CYCLES], ed_pair[CYCLES][CYCLES];
  {
    ef_dp_req q[CYCLES + CYCLES * CYCLES]; ef_dp_res rs[CYCLES + CYCLES * CYCLES]; size_t nq = 0;
    int sp[CYCLES], sx[CYCLES][CYCLES];
    for (int i = 0; i < CYCLES; ++i) {
      sp[i] = -1;
      if (variant1 && cut_factor[i] != NULL) {
        const ef_dp_req x = { EF_DP_ED, cut_factor[i], strlen(cut_factor[i]), prev_match[i], strlen(prev_match[i]), 0, 0, 0, 0, 1 };
        sp[i] = (int)nq; q[nq++] = x;
      }
      for (int j = 0; j < CYCLES; ++j) {
        sx[i][j] = -1;
        const char *a = NULL, *b = NULL;
        if (ext_cut[i] != NULL && ext_match[j] != NULL) { a = ext_cut[i]; b = ext_match[j]; }
        else if (cut_factor[i] != NULL && match_str[j] != NULL) { a = cut_factor[i]; b = match_str[j]; }
        if (a) { const ef_dp_req x = { EF_DP_ED, a, strlen(a), b, strlen(b), 0, 0, 0, 0, 1 }; sx[i][j] = (int)nq; q[nq++] = x; }
      }
    }
    if (ef_dp_many(be, q, rs, nq) != 0) { fprintf(stderr, "* FATAL edit-distance backend failed\n"); abort(); }
    for (int i = 0; i < CYCLES; ++i) {
      ed_prev[i] = sp[i] >= 0 ? (uint32_t)rs[sp[i]].v[0] : 0;
      for (int j = 0; j < CYCLES; ++j) ed_pair[i][j] = sx[i][j] >= 0 ? (uint32_t)rs[sx[i][j]].v[0] : 0;
    }
  }
  bool stop = false;
  if (variant1) {
    unsigned error = 1000, edit_prev = 1000;
    for (int i = 0; i < CYCLES && !stop; ++i) {
      for (int j = 0; j < CYCLES && !stop; ++j) {
        if (cut_factor[i] != NULL && match_str[j] != NULL) {
          edit_prev = ed_prev[i];
          if (edit_prev <= 5) {
            if (ext_cut[i] != NULL && ext_match[j] != NULL)
              error = ed_pair[i][j] - edit_prev - (unsigned)ext_error;
            else
              error = ed_pair[i][j] - edit_prev;
          }
        }
        if (error <= 1) {
          if (r2l) {
            *out_factor_left = al->new_acceptor_factor_left + est_cut[i];
            *out_donor_right = al->new_donor_right_on_gen + sub_dim[j];
            *out_acc_left = al->new_acceptor_left_on_gen + gen_cut[i];
          } else {
            *out_factor_left = al->new_acceptor_factor_left - est_cut[i];
            *out_donor_right = al->new_donor_right_on_gen - gen_cut[i];
            *out_acc_left = al->new_acceptor_left_on_gen - sub_dim[j];
          }
          stop = true;
        }
      }
    }
  } else {
    int error = 1000, edit = 1000;
    for (int i = 0; i < CYCLES && !stop; ++i) {
      for (int j = 0; j < CYCLES && !stop; ++j) {
        if (ext_cut[i] != NULL && ext_match[j] != NULL) edit = (int)ed_pair[i][j] - ext_error;
        else if (cut_factor[i] != NULL && match_str[j] != NULL) edit = (int)ed_pair[i][j];
        else edit = 1000;
        if (edit < error) {
          error = edit;
          if (r2l) {
            *out_factor_left = al->new_acceptor_factor_left + est_cut[i];
            *out_donor_right = al->new_donor_right_on_gen + sub_dim[j];
            *out_acc_left = al->new_acceptor_left_on_gen + gen_cut[i];
          } else {
            *out_factor_left = al->new_acceptor_factor_left - est_cut[i];
            *out_donor_right = al->new_donor_right_on_gen - gen_cut[i];
            *out_acc_left = al->new_acceptor_left_on_gen - sub_dim[j];
          }
        }
        if (error == 0) stop = true;
      }
    }
  }
  for (int i = 0; i < CYCLES; ++i) { sp_put(sp, cut_factor[i]); sp_put(sp, match_str[i]); sp_put(sp, prev_match[i]); sp_put(sp, ext_cut[i]); sp_put(sp, ext_match[i]); }
  return stop;
}

/* Try_Burset_after_match (:267-344) */
static void try_burset_after_match(const char* est, const char* gen, int* factor_left, int* donor_right,
                                   int* acc_left, int donor_factor_left, int acc_factor_right) {
  int sf = *factor_left, sa = *acc_left, sd = *donor_right;
  int uf = sf, ua = sa, ud = sd;
  int frequency = 0;
  bool right_to_left = false, stop = false;
  while ((!stop && est[sf] == gen[sa]) && sf > donor_factor_left + 1) {
    if (sf == 0 || sd == -1) stop = true;
    else {
      const int f = ef_check_burset_patterns(gen, sd, sa);
      if (f > frequency) { frequency = f; uf = sf; ua = sa; ud = sd; }
      --sf; --sd; --sa;
    }
  }
  sf = *factor_left; sa = *acc_left + 1; sd = *donor_right + 1;
  stop = false;
  const size_t el = strlen(est), gl = ef_genomic_len(gen);
  while ((!stop && est[sf] == gen[sd]) && sf < acc_factor_right) {
    if ((unsigned)sf == el || (unsigned)sa == gl) stop = true;
    else {
      const int f = ef_check_burset_patterns(gen, sd, sa);
      if (f > frequency) { frequency = f; uf = sf; ua = sa; ud = sd; right_to_left = true; }
      ++sf; ++sd; ++sa;
    }
  }
  if (right_to_left) uf += 1;
  *factor_left = uf; *donor_right = ud; *acc_left = ua;
}

/* appends real_substring(index, length, s) (src/util.c:138-158) to dst[n..]; returns the new length */
static size_t append_substring(char* dst, size_t n, int index, int length, const char* s) {
  if (index < 0) { length += index; index = 0; }
  const char* p = s + index;
  for (int k = 0; k < length && p[k] != '\0'; ++k) dst[n++] = p[k];
  return n;
}

/* The two strings of the gap alignment of one intron (:60-116): concatenations of real_substring() pieces --
 * donor suffix + unaligned EST gap + acceptor prefix on the EST, and donor suffix + intron prefix + intron
 * suffix + acceptor prefix on the genomic sequence -- appended straight into two buffers (same clamping, same
 * stop at the terminator). */
void ef_gap_window_build(const ef_config* cfg, const ef_seq* gen_info, const ef_seq* est_info, const ef_factor* donor,
                         const ef_factor* acceptor, ef_gap_window* w) {
  const int sp_est = cfg->suffpref_length_on_est, sp_int = cfg->suffpref_length_for_intron, sp_gen = cfg->suffpref_length_on_gen;
  const char* G = gen_info->seq;
  const char* E = est_info->seq;
  int dsl_gen = donor->GEN_start;
  if (donor->GEN_end - sp_gen + 1 >= dsl_gen) dsl_gen = donor->GEN_end - sp_gen + 1;
  int dsl_est = donor->EST_start;
  if (donor->EST_end - sp_est + 1 >= dsl_est) dsl_est = donor->EST_end - sp_est + 1;
  int apr_gen = acceptor->GEN_end;
  if (acceptor->GEN_start + sp_gen - 1 <= apr_gen) apr_gen = acceptor->GEN_start + sp_gen - 1;
  int apr_est = acceptor->EST_end;
  if (acceptor->EST_start + sp_est - 1 <= apr_est) apr_est = acceptor->EST_start + sp_est - 1;
  const bool has_gap = donor->EST_end != acceptor->EST_start - 1;
  const int piece_e[3][2] = { { dsl_est, donor->EST_end - dsl_est + 1 },
                              { donor->EST_end + 1, has_gap ? acceptor->EST_start - donor->EST_end - 1 : 0 },
                              { acceptor->EST_start, apr_est - acceptor->EST_start + 1 } };
  const int piece_g[4][2] = { { dsl_gen, donor->GEN_end - dsl_gen + 1 }, { donor->GEN_end + 1, sp_int },
                              { acceptor->GEN_start - sp_int, sp_int }, { acceptor->GEN_start, apr_gen - acceptor->GEN_start + 1 } };
  size_t cap_e = 1, cap_g = 1;
  for (int k = 0; k < 3; ++k) cap_e += piece_e[k][1] > 0 ? (size_t)piece_e[k][1] : 0;
  for (int k = 0; k < 4; ++k) cap_g += piece_g[k][1] > 0 ? (size_t)piece_g[k][1] : 0;
  w->seq_est = cap_e <= sizeof w->buf_e ? w->buf_e : (char*)malloc(cap_e);
  w->seq_gen = cap_g <= sizeof w->buf_g ? w->buf_g : (char*)malloc(cap_g);
  size_t le = 0, lg = 0;
  for (int k = 0; k < 3; ++k) le = append_substring(w->seq_est, le, piece_e[k][0], piece_e[k][1], E);
  for (int k = 0; k < 4; ++k) lg = append_substring(w->seq_gen, lg, piece_g[k][0], piece_g[k][1], G);
  w->seq_est[le] = '\0'; w->seq_gen[lg] = '\0';
  w->le = le; w->lg = lg; w->dsl_est = dsl_est; w->dsl_gen = dsl_gen;
  w->deleted_intron_dim = acceptor->GEN_start - donor->GEN_end - 1 - 2 * sp_int;
}
void ef_gap_window_release(ef_gap_window* w) {
  if (w->seq_est != w->buf_e) free(w->seq_est);
  if (w->seq_gen != w->buf_g) free(w->seq_gen);
  w->seq_est = w->seq_gen = NULL;
}

/* refine_intron (:47-265).  `ahead` (may be NULL): a gap alignment asked before its turn -- its two strings and
 * the answer; it is taken when the strings of this intron, built from the exons as they are NOW, are the same. */
bool ef_refine_intron(const ef_config* cfg, const ef_seq* gen_info, const ef_seq* est_info, ef_factor* donor,
                      ef_factor* acceptor, bool first_intron, ef_backend* be, ef_gap_ahead* ahead) {
  const char* G = gen_info->seq;
  const char* E = est_info->seq;
  ef_gap_window w;
  w.seq_est = w.seq_gen = NULL;
  ef_dp_res rs;
  memset(&rs, 0, sizeof rs);
  int dsl_est, dsl_gen, deleted_intron_dim;
  /* the same two exons as when the alignment was asked: the same strings, nothing to build */
  const bool same_exons = ahead && ahead->res.s0 && memcmp(&ahead->donor, donor, sizeof *donor) == 0 &&
                          memcmp(&ahead->acceptor, acceptor, sizeof *acceptor) == 0;
  if (same_exons) {
    dsl_est = ahead->w.dsl_est; dsl_gen = ahead->w.dsl_gen; deleted_intron_dim = ahead->w.deleted_intron_dim;
    rs = ahead->res;                             /* the rows are ours now */
    ahead->res.s0 = ahead->res.s1 = NULL;
    if (ef_prof_on) ++ef_prof.ahead_hits;
  } else {
    ef_gap_window_build(cfg, gen_info, est_info, donor, acceptor, &w);
    dsl_est = w.dsl_est; dsl_gen = w.dsl_gen; deleted_intron_dim = w.deleted_intron_dim;
    if (ahead && ahead->res.s0 && ahead->w.le == w.le && ahead->w.lg == w.lg &&
        memcmp(ahead->w.seq_est, w.seq_est, w.le) == 0 && memcmp(ahead->w.seq_gen, w.seq_gen, w.lg) == 0) {
      rs = ahead->res;
      ahead->res.s0 = ahead->res.s1 = NULL;
      if (ef_prof_on) ++ef_prof.ahead_hits;
    } else {
      ef_dp_req rq = { EF_DP_GAP, w.seq_est, w.le, w.seq_gen, w.lg, 0, 0, 0, 0, 1 };
      if (be->dp(be->self, &rq, &rs) != 0) { fprintf(stderr, "* FATAL gap alignment backend failed\n"); abort(); }
    }
  }
  gap_aln al;
  al.est_row = rs.s0; al.gen_row = rs.s1;      /* one zero-padded block (ef_dp_res), released at `done` */
  al.dim = rs.v[0]; al.factor_cut = rs.v[1]; al.intron_start = rs.v[2]; al.intron_end = rs.v[3];
  al.intron_start_on_align = rs.v[4]; al.intron_end_on_align = rs.v[5];
  al.new_acceptor_factor_left = dsl_est + al.factor_cut;
  al.new_donor_right_on_gen = dsl_gen + al.intron_start - 1;
  al.new_acceptor_left_on_gen = dsl_gen + al.intron_end + deleted_intron_dim + 1;
  if (w.seq_est) ef_gap_window_release(&w);

  bool result = false;
  if (al.new_acceptor_factor_left == donor->EST_start) {
    if (first_intron) {
      acceptor->EST_start = al.new_acceptor_factor_left;
      acceptor->GEN_start = al.new_acceptor_left_on_gen;
      result = true;
    }
    goto done;
  }
  if (al.new_acceptor_left_on_gen - al.new_donor_right_on_gen < cfg->min_intron_length) goto done;
  {
    const int dshift = abs(al.new_donor_right_on_gen - donor->GEN_end);
    const int ashift = abs(al.new_acceptor_left_on_gen - acceptor->GEN_start);
    if (dshift > 20 || ashift > 20) goto done;
  }
  {
    int lc = 0, lg = 0, le2 = 0, rc = 0, rg = 0, re = 0;
    find_before_left(&al, al.intron_start_on_align - 1, &lc, &lg, &le2, "GT");
    find_AG_after_right(&al, al.intron_end_on_align + 1, &rc, &rg, &re);
    int fin_d, fin_a, fin_f;
    if (lg == 0 && rg == 0) {
      fin_d = al.new_donor_right_on_gen; fin_a = al.new_acceptor_left_on_gen; fin_f = al.new_acceptor_factor_left;
    } else {
      int sd = 0, sa = 0, sf = 0;
      if (!shift_generic(E, G, &al, true, true, "GT", &sd, &sa, &sf, be)) {
        sd = sa = sf = 0;
        if (!shift_generic(E, G, &al, false, true, "GT", &sd, &sa, &sf, be)) {
          sd = sa = sf = 0;
          if (!shift_generic(E, G, &al, true, false, "GC", &sd, &sa, &sf, be)) {
            sd = sa = sf = 0;
            if (!shift_generic(E, G, &al, false, false, "GC", &sd, &sa, &sf, be)) {
              sf = al.new_acceptor_factor_left; sd = al.new_donor_right_on_gen; sa = al.new_acceptor_left_on_gen;
              try_burset_after_match(E, G, &sf, &sd, &sa, donor->EST_start, acceptor->EST_end);
            }
          }
        }
      }
      fin_d = sd; fin_a = sa; fin_f = sf;
      if (fin_a > acceptor->GEN_end || fin_d < donor->GEN_start) goto done;
    }
    donor->GEN_end = fin_d;
    acceptor->GEN_start = fin_a;
    acceptor->EST_start = fin_f;
    donor->EST_end = acceptor->EST_start - 1;
    result = true;
  }
done:
  free(al.est_row);                          /* est_row heads the block of both rows */
  return result;
}
