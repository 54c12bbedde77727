/* Intron classification U12 / U2 / not classified by position weight matrices (MatInspector
 * score) and branch-point search.  Behaviour follows src/classify-intron.c:95-229,535-663,
 * 1498-1553 of the reference; only the class is needed on the est-fact path, which depends on the
 * branch-point matrices and the four 5' splice-site matrices (the 3' matrices only feed score3,
 * unused here).  IEEE double arithmetic in the reference's operation order; the matrices below are
 * the reference's DATA (GetPWMfor* at :665-1100), each entry + 0.00001f as there. */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include "estfact.h"

#define N_PWM 6
static const int pwm_len[N_PWM] = { 12, 12, 14, 14, 13, 14 };

static const double pwm_raw_0[4][12] = {  /* GetPWMforBPS_9 */
  { 0.16, 0.19, 0.08, 0.09, 0.01, 0.01, 0.01, 0.01, 1.00, 0.94, 0.01, 0.28 },
  { 0.15, 0.18, 0.12, 0.09, 0.90, 0.90, 0.01, 0.01, 0.00, 0.01, 0.84, 0.16 },
  { 0.18, 0.14, 0.13, 0.07, 0.01, 0.01, 0.01, 0.01, 0.00, 0.04, 0.01, 0.04 },
  { 0.51, 0.49, 0.67, 0.75, 0.08, 0.08, 0.97, 0.97, 0.00, 0.01, 0.14, 0.52 },
};
static const double pwm_raw_1[4][12] = {  /* GetPWMforBPS_10 */
  { 0.12, 0.12, 0.13, 0.09, 0.01, 0.01, 0.01, 0.01, 0.70, 1.00, 0.02, 0.23 },
  { 0.15, 0.17, 0.20, 0.10, 0.86, 0.92, 0.03, 0.01, 0.02, 0.00, 0.82, 0.25 },
  { 0.21, 0.18, 0.12, 0.05, 0.01, 0.01, 0.01, 0.03, 0.24, 0.00, 0.02, 0.05 },
  { 0.52, 0.53, 0.55, 0.76, 0.12, 0.06, 0.95, 0.95, 0.04, 0.00, 0.14, 0.47 },
};
static const double pwm_raw_2[4][14] = {  /* GetPWMfor5PrimeGTAGU12 */
  { 0.293478260869565, 0.271739130434783, 0.217391304347826, 0.00, 0.00, 0.983695652173913, 0.00543478260869565, 0.00543478260869565, 0.0217391304347826, 0.0108695652173913, 0.0380434782608696, 0.103260869565217, 0.184782608695652, 0.40 },
  { 0.239130434782609, 0.326086956521739, 0.184782608695652, 0.00, 0.00, 0.00543478260869565, 0.00543478260869565, 0.983695652173913, 0.907608695652174, 0.016304347826087, 0.0489130434782609, 0.195652173913043, 0.266304347826087, 0.20 },
  { 0.239130434782609, 0.152173913043478, 0.0543478260869565, 1.00, 0.00, 0.00543478260869565, 0.00543478260869565, 0.00543478260869565, 0.00543478260869565, 0.0108695652173913, 0.0489130434782609, 0.0869565217391304, 0.141304347826087, 0.20 },
  { 0.228260869565217, 0.25, 0.543478260869565, 0.00, 1.00, 0.00543478260869565, 0.983695652173913, 0.00543478260869565, 0.0652173913043478, 0.96195652173913, 0.864130434782609, 0.614130434782609, 0.407608695652174, 0.20 },
};
static const double pwm_raw_3[4][14] = {  /* GetPWMfor5PrimeATACU12 */
  { 0.271028037383178, 0.280373831775701, 0.299065420560748, 1.00, 0.00, 0.97196261682243, 0.00934579439252336, 0.00934579439252336, 0.00934579439252336, 0.0186915887850467, 0.0186915887850467, 0.0467289719626168, 0.177570093457944, 0.40 },
  { 0.280373831775701, 0.271028037383178, 0.289719626168224, 0.00, 0.00, 0.00934579439252336, 0.00934579439252336, 0.97196261682243, 0.962616822429907, 0.0186915887850467, 0.0747663551401869, 0.205607476635514, 0.224299065420561, 0.20 },
  { 0.224299065420561, 0.149532710280374, 0.299065420560748, 0.00, 0.00, 0.00934579439252336, 0.00934579439252336, 0.00934579439252336, 0.00934579439252336, 0.00934579439252336, 0.0186915887850467, 0.0373831775700935, 0.233644859813084, 0.20 },
  { 0.224299065420561, 0.299065420560748, 0.11214953271028, 0.00, 1.00, 0.00934579439252336, 0.97196261682243, 0.00934579439252336, 0.0186915887850467, 0.953271028037383, 0.88785046728972, 0.710280373831776, 0.364485981308411, 0.20 },
};
static const double pwm_raw_4[4][13] = {  /* GetPWMfor5PrimeGTAGU2 */
  { 0.341467901547831, 0.660619132199949, 0.0973420451662015, 0.00, 0.00, 0.596517381375285, 0.763251712763258, 0.0781781273788379, 0.183893681806648, 0.297487947221517, 0.224714539456991, 0.222602131438721, 0.225862725196651 },
  { 0.375069779243847, 0.0834496320730779, 0.00990230905861456, 0.00, 0.00, 0.0225640700329866, 0.0241055569652372, 0.0350291804110632, 0.144626998223801, 0.189672671910683, 0.247881248414108, 0.259737376300431, 0.23369068764273 },
  { 0.183646282669373, 0.10822126363867, 0.840998477543771, 1.00, 0.00, 0.369373255518904, 0.11949378330373, 0.830106571936057, 0.188219994925146, 0.304884547069272, 0.23943161634103, 0.245045673686882, 0.258709718345598 },
  { 0.0998160365389495, 0.147709972088302, 0.0517571682314133, 0.00, 1.00, 0.0115452930728242, 0.0931489469677747, 0.0566861202740421, 0.483259325044405, 0.207954833798528, 0.287972595787871, 0.272614818573966, 0.281736868815022 },
};
static const double pwm_raw_5[4][14] = {  /* GetPWMfor5PrimeGCAGU2 */
  { 0.402203856749311, 0.873278236914601, 0.0199724517906336, 0.00, 0.00, 0.924931129476584, 0.831955922865014, 0.00619834710743802, 0.075068870523416, 0.330578512396694, 0.192148760330579, 0.194214876033058, 0.245179063360882, 0.40 },
  { 0.368457300275482, 0.0130853994490358, 0.00206611570247934, 0.00, 1.00, 0.0137741046831956, 0.0254820936639118, 0.00206611570247934, 0.0867768595041322, 0.158402203856749, 0.272038567493113, 0.305785123966942, 0.214187327823691, 0.20 },
  { 0.172176308539945, 0.0433884297520661, 0.974517906336088, 1.00, 0.00, 0.0564738292011019, 0.087465564738292, 0.988292011019284, 0.0929752066115702, 0.34228650137741, 0.213498622589532, 0.221763085399449, 0.260330578512397, 0.20 },
  { 0.0571625344352617, 0.0702479338842975, 0.0034435261707989, 0.00, 0.00, 0.00482093663911846, 0.0550964187327824, 0.0034435261707989, 0.745179063360882, 0.168732782369146, 0.322314049586777, 0.278236914600551, 0.28030303030303, 0.20 },
};

static double* PWM[N_PWM];   /* [base*len + pos] */
static double* CV[N_PWM];
static double* MAXV[N_PWM];

static void load_one(int k, const double* raw) {
  const int n = pwm_len[k];
  PWM[k] = (double*)malloc(4 * n * sizeof(double));
  for (int i = 0; i < 4; ++i) for (int j = 0; j < n; ++j) PWM[k][i * n + j] = raw[i * n + j] + 0.00001f;
  CV[k] = (double*)malloc(n * sizeof(double));        /* GetCVectorForPWM (:1498-1518) */
  MAXV[k] = (double*)malloc(n * sizeof(double));      /* GetMAXVectorForPWM (:1520-1537) */
  for (int i = 0; i < n; ++i) {
    CV[k][i] = 0;
    for (int j = 0; j < 4; ++j) CV[k][i] += PWM[k][j * n + i] * log(PWM[k][j * n + i]);
    CV[k][i] += log(5.0f);
    CV[k][i] *= (100.0f / log(5.0f));
    MAXV[k][i] = 0.0f;
    for (int j = 0; j < 4; ++j) if (PWM[k][j * n + i] > MAXV[k][i]) MAXV[k][i] = PWM[k][j * n + i];
  }
}

static void load_all(void) {
  static int loaded = 0;
  if (loaded) return;

  load_one(0, &pwm_raw_0[0][0]);
  load_one(1, &pwm_raw_1[0][0]);
  load_one(2, &pwm_raw_2[0][0]);
  load_one(3, &pwm_raw_3[0][0]);
  load_one(4, &pwm_raw_4[0][0]);
  load_one(5, &pwm_raw_5[0][0]);
  loaded = 1;
}

void ef_classify_init(void) { load_all(); }

/* GetMatInspectorScoreOfaMotif (:620-663).  A character outside ACGTN indexes row -1 in the
 * reference (out of bounds under NDEBUG); we return a score no motif can reach instead. */
static double motif_score(const char* s, int k) {
  const int n = pwm_len[k];
  double den = 0.0f, num = 0.0f;
  for (int i = 0; i < n; ++i) {
    int idx = -1;
    switch (s[i]) {
      case 'N': case 'n': case 'A': case 'a': idx = 0; break;
      case 'C': case 'c': idx = 1; break;
      case 'G': case 'g': idx = 2; break;
      case 'T': case 't': idx = 3; break;
    }
    if (idx < 0) return -1.0;
    num += CV[k][i] * PWM[k][idx * n + i];
    den += CV[k][i] * MAXV[k][i];
  }
  return num / den;
}

/* SearchBPSinIntronSequenceWithMathInspector (:575-618) over intron[0..length) */
static int search_bps(const char* intron, size_t length, int k, double* score, int range_start, int range_end) {
  *score = 0.0f;
  if (length < (unsigned)range_start) return -1;
  int start_w = (int)length - range_end, end_w = (int)length - range_start;
  if (start_w < 0) start_w = 0;
  int start_bps = -1;
  bool first = true;
  char win[13];
  for (int i = start_w; i <= end_w; ++i) {
    /* real_substring(i, 12, intronSequence): NUL-padded when the window passes the end */
    for (int c = 0; c < 12; ++c) win[c] = (size_t)(i + c) < length ? intron[i + c] : '\0';
    win[12] = '\0';
    const double sc = motif_score(win, k);
    if (first || sc >= *score) { *score = sc; start_bps = i; first = false; }
  }
  return start_bps;
}

/* ExistsGoodBPSinIntronSequenceWithMathInspector (:535-573) */
static int good_bps(const char* intron, size_t length, int range_start, int range_end) {
  if (range_end > (int)length) return -1;
  double s9 = 0.0f, s10 = 0.0f;
  const int b9 = search_bps(intron, length, 0, &s9, range_start, range_end);
  const int b10 = search_bps(intron, length, 1, &s10, range_start, range_end);
  if (s9 > s10) return s9 > 0.75f ? b9 : -1;
  return s10 > 0.75f ? b10 : -1;
}

static double score5(const char* gen, int splice5, int k) {           /* GetScoreOf5Prime*BySS */
  char win[24];
  int idx = splice5 - 3, len = pwm_len[k];
  if (idx < 0) { len += idx; idx = 0; }                     /* real_substring clamps a negative index */
  if (len < 0) len = 0;
  int c = 0;
  for (; c < len && gen[idx + c] != '\0'; ++c) win[c] = gen[idx + c];
  for (; c < 24; ++c) win[c] = '\0';
  return motif_score(win, k);
}

/* ---- per-gene tables ------------------------------------------------------------------------------
 * bps_memo[E]   verdict of ExistsGoodBPS... for an intron that ends at E (exclusive) and is at least
 *               30 long: the scan looks at the windows that start 30..14 bases before E only, all of
 *               them inside the intron, so the verdict is a property of E.
 * score5_tab    score5(gen, start, k) for the four 5' matrices k = 2..5 at every start.
 * Both are filled with the very functions the per-intron path calls, over ranges of positions on a
 * few threads. */
typedef struct { ef_seq* gs; size_t lo, hi, n; } prep_range;

/* the verdict of classify_genomic_intron_start_end's last lines (:209-228) for given scores */
static inline unsigned cmp_bits(double u12, double u2) {
  return (u12 > u2 ? 1u : 0u) | ((u12 - u2 > 0.25 && u12 >= 0.75) ? 2u : 0u);
}
enum { P5_GT = 1, P5_GC = 2, P5_AT = 3, P3_AG = 1, P3_AC = 2 };
static inline int pair_is(const char* p, char a, char b) {       /* strcmp with "xy" or "XY" */
  return (p[0] == a && p[1] == b) || (p[0] == (char)(a + 32) && p[1] == (char)(b + 32));
}

static void* prepare_range(void* arg) {
  prep_range* r = (prep_range*)arg;
  const char* gen = r->gs->seq;
  for (size_t e = r->lo; e < r->hi; ++e) {
    if (e >= 30) r->gs->bps_memo[e] = good_bps(gen + (e - 30), 30, 14, 30) != -1 ? 2 : 1;
    double sc[4];
    for (int k = 0; k < 4; ++k) sc[k] = r->gs->score5_tab[k][e] = score5(gen, (int)e, 2 + k);
    /* cls_start[e]: bits 0-1 the kind of gen[e], gen[e+1]; bits 2-3 cmp_bits of the scores that kind selects when
     * the intron's end agrees (GT..AG, GC..AG, AT..AC); bits 4-5 cmp_bits of the general case */
    unsigned char cs = 0;
    if (e + 1 < r->n) {
      const int kind = pair_is(gen + e, 'G', 'T') ? P5_GT : pair_is(gen + e, 'G', 'C') ? P5_GC : pair_is(gen + e, 'A', 'T') ? P5_AT : 0;
      const double m23 = sc[1] > sc[0] ? sc[1] : sc[0];          /* u12 = s2, then "if (t > u12) u12 = t" with t = s3 */
      const double m45 = sc[3] > sc[2] ? sc[3] : sc[2];
      unsigned own = 0;
      if (kind == P5_GT) own = cmp_bits(sc[0], sc[2]);
      else if (kind == P5_GC) own = cmp_bits(m23, sc[3]);
      else if (kind == P5_AT) own = cmp_bits(sc[1], m45);
      cs = (unsigned char)((unsigned)kind | (own << 2) | (cmp_bits(m23, m45) << 4) | 0x80u);     /* bit 7: filled */
    }
    r->gs->cls_start[e] = cs;
    /* cls_end[e] for an intron whose LAST character is gen[e]: bit 0-1 kind of gen[e-1], gen[e]; bit 2 branch
     * point found (the verdict bps_memo keeps under e + 1, filled by the iteration of e + 1 or below) */
    unsigned char ce = 0;
    if (e >= 1 && e < r->n) ce = (unsigned char)((pair_is(gen + e - 1, 'A', 'G') ? P3_AG : pair_is(gen + e - 1, 'A', 'C') ? P3_AC : 0) | 0x80u);
    r->gs->cls_end[e] = ce;
  }
  return NULL;
}

void ef_classify_prepare(ef_seq* gs) {
  load_all();
  const size_t n = strlen(gs->seq);
  if (!gs->bps_memo) gs->bps_memo = (unsigned char*)calloc(n + 2, 1);
  for (int k = 0; k < 4; ++k) { free(gs->score5_tab[k]); gs->score5_tab[k] = (double*)malloc((n + 2) * sizeof(double)); }
  free(gs->cls_start); free(gs->cls_end);
  gs->cls_start = (unsigned char*)calloc(n + 2, 1); gs->cls_end = (unsigned char*)calloc(n + 2, 1);
  gs->score5_len = n + 1;
  enum { T = 8 };
  prep_range rg[T]; pthread_t th[T]; bool started[T];
  for (int t = 0; t < T; ++t) {
    rg[t].gs = gs; rg[t].n = n; rg[t].lo = (n + 1) * (size_t)t / T; rg[t].hi = (n + 1) * (size_t)(t + 1) / T;
    started[t] = n >= 4096 && pthread_create(&th[t], NULL, prepare_range, &rg[t]) == 0;
    if (!started[t]) prepare_range(&rg[t]);
  }
  for (int t = 0; t < T; ++t) if (started[t]) pthread_join(th[t], NULL);
  /* the branch-point verdict of the intron that ends ON e is kept under e + 1 */
  for (size_t e = 0; e + 1 <= n; ++e) if (gs->bps_memo[e + 1] == 2) gs->cls_end[e] |= 4u;
}

static inline double score5_at(const ef_seq* gs, int start, int k) {
  if (gs->score5_tab[0] && start >= 0 && (size_t)start < gs->score5_len) return gs->score5_tab[k - 2][start];
  return score5(gs->seq, start, k);
}

/* classify_genomic_intron_start_end (:95-229), class only.  The reference copies the intron
 * (real_substring); we read it in place: intron = gen[start .. start+il) */
static int classify_uncached(const ef_seq* gs, int start, int end);

/* The class depends only on (genomic, start, end) and the ESTs of a gene keep proposing the same
 * few introns, so each thread remembers recent answers (direct-mapped; the genomic sequence is
 * immutable for the whole run). */
int ef_classify_intron(const ef_seq* gs, int start, int end) {
  const char* gen = gs->seq;
  /* the common case from the two byte tables: an intron of at least 30 characters inside the sequence */
  if (gs->cls_start && start >= 0 && end - start >= 29 && (size_t)end < gs->score5_len - 1) {
    const unsigned cs = gs->cls_start[start], ce = gs->cls_end[end];
    if ((cs & ce & 0x80u) != 0) {
      const unsigned p5 = cs & 3u, p3 = ce & 3u;
      const bool own = (p5 == P5_GT && p3 == P3_AG) || (p5 == P5_GC && p3 == P3_AG) || (p5 == P5_AT && p3 == P3_AC);
      const unsigned bits = own ? (cs >> 2) & 3u : (cs >> 4) & 3u;
      if (ce & 4u) return (bits & 1u) ? 0 : 1;                    /* branch point: u12 > u2 ? U12 : U2 */
      if (own && p5 != P5_AT) return 1;                           /* GT..AG / GC..AG without one: U2 */
      return (bits & 2u) ? 0 : 2;
    }
  }
  typedef struct { const char* gen; unsigned epoch; int start, end, type; } slot;
  static _Thread_local slot memo[1024];
  const unsigned ep = ef_genomic_epoch_now();       /* (address, epoch): see ef_genomic_len */
  slot* m = &memo[((uint32_t)start * 2654435761u ^ (uint32_t)end * 40503u) >> 7 & 1023u];
  if (m->gen == gen && m->epoch == ep && m->start == start && m->end == end) return m->type;
  const int type = classify_uncached(gs, start, end);
  m->gen = gen; m->epoch = ep; m->start = start; m->end = end; m->type = type;
  return type;
}

static int classify_uncached(const ef_seq* gs, int start, int end) {
  const char* gen = gs->seq;
  load_all();
  int idx = start, want = end - start + 1;
  if (idx < 0) { want += idx; idx = 0; }
  if (want < 0) want = 0;
  const char* intron = gen + idx;
  /* real_substring stops at the terminator of the sequence: the intron is cut at the end of the genomic
   * sequence, whose length is known (looking for the terminator meant reading the whole intron, up to
   * tens of kilobases, for every candidate of the small-exon search) */
  const size_t gl = ef_genomic_len(gen);
  const size_t il = (size_t)idx >= gl ? 0 : ((size_t)want < gl - (size_t)idx ? (size_t)want : gl - (size_t)idx);
  /* The branch-point scan looks at the windows 30..14 bases before the END of the intron only (and
   * gives up on introns shorter than 30), so its verdict is a property of the end position; the
   * small-exon search proposes thousands of introns sharing a few ends. */
  int bps;
  if (il < 30 || !gs->bps_memo) bps = good_bps(intron, il, 14, 30);
  else {
    unsigned char* m = &gs->bps_memo[(size_t)idx + il];
    if (*m == 0) *m = good_bps(intron, il, 14, 30) != -1 ? 2 : 1;    /* same value whoever writes it */
    bps = *m == 2 ? 0 : -1;                                             /* only "found or not" is used below */
  }
  char p5[3] = { 0, 0, 0 }, p3[3] = { 0, 0, 0 };
  for (size_t c = 0; c < 2 && c < il; ++c) p5[c] = intron[c];
  if (il >= 2) { p3[0] = intron[il - 2]; p3[1] = intron[il - 1]; }
  else for (size_t c = 0; c < il; ++c) p3[c] = intron[c];
  double u12 = 0.0f, u2 = 0.0f, t;
  int pt_type = 1;
  const bool ag = !strcmp(p3, "ag") || !strcmp(p3, "AG");
  if ((!strcmp(p5, "gt") || !strcmp(p5, "GT")) && ag) {
    pt_type = 0;
    u12 = score5_at(gs, start, 2);
    u2 = score5_at(gs, start, 4);
  } else if ((!strcmp(p5, "gc") || !strcmp(p5, "GC")) && ag) {
    pt_type = 0;
    u2 = score5_at(gs, start, 5);
    u12 = score5_at(gs, start, 2);
    t = score5_at(gs, start, 3); if (t > u12) u12 = t;
  } else if ((!strcmp(p5, "at") || !strcmp(p5, "AT")) && (!strcmp(p3, "ac") || !strcmp(p3, "AC"))) {
    u12 = score5_at(gs, start, 3);
    u2 = score5_at(gs, start, 4);
    t = score5_at(gs, start, 5); if (t > u2) u2 = t;
  } else {
    u12 = score5_at(gs, start, 2);
    t = score5_at(gs, start, 3); if (t > u12) u12 = t;
    u2 = score5_at(gs, start, 4);
    t = score5_at(gs, start, 5); if (t > u2) u2 = t;
  }
  int type = 2;
  if (bps != -1) type = u12 > u2 ? 0 : 1;
  else if (pt_type == 0) type = 1;
  else if (u12 - u2 > 0.25 && u12 >= 0.75) type = 0;
  return type;
}
