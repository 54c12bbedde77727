"""DP test cases shared by the parity tests, smoke() and the golden-vector tools.

A case = (kind, a, b, params).  `expected(O)` evaluates it with the CPU oracle
(tests/oracle_lib.py); `add_to(joblist)` queues it for the HIP library; `check_case` compares."""
import random

ALIGN, GAP, ED, KBAND, LCF, BORDERS, AFFIX = range(7)
KIND_NAMES = ["ALIGN", "GAP", "ED", "KBAND", "LCF", "BORDERS", "AFFIX"]

FIELDS = {
    ALIGN: ("score", "dim", "ea", "ga"),
    GAP: ("dim", "factor_cut", "intron_start", "intron_end", "intron_start_on_align",
          "intron_end_on_align", "ea", "ga"),
    ED: ("score",),
    KBAND: ("ok", "edit"),
    LCF: ("len", "occ1", "occ2"),
    BORDERS: ("ok", "off_p", "off_t1", "off_t2", "ed"),
    AFFIX: ("valid", "ecut", "gcut"),
}


class Case:
    def __init__(self, kind, a, b, p0=0, p1=0, p2=0, b_tail=b""):
        self.kind, self.a, self.b = kind, a, b
        self.p0, self.p1, self.p2, self.b_tail = p0, p1, p2, b_tail

    def __repr__(self):
        return "Case(%s, a=%r, b=%r, p=(%d,%d,%d), tail=%r)" % (
            KIND_NAMES[self.kind], self.a[:60], self.b[:60], self.p0, self.p1, self.p2, self.b_tail)

    def add_to(self, jl):
        return jl.add(self.kind, self.a, self.b, self.p0, self.p1, self.p2, self.b_tail)

    def expected(self, O):
        k = self.kind
        if k == ALIGN:
            return O.align(self.a, self.b)
        if k == GAP:
            return O.gap_align(self.a, self.b)
        if k == ED:
            return dict(score=O.edit_distance(self.a, self.b))
        if k == KBAND:
            return O.kband(self.a, self.b, self.p0)
        if k == LCF:
            return O.lcf(self.a, self.b)
        if k == BORDERS:
            return O.refine_borders(self.a, self.b, self.p0, self.p1, self.p2, self.b_tail)
        if k == AFFIX:
            r = O.longest_affix(self.a, self.b)
            if not r["valid"]:
                r["ecut"] = r["gcut"] = 0
            return r
        raise ValueError(k)


def check_case(case, got, O, expected=None):
    exp = expected if expected is not None else case.expected(O)
    if got.get("status", 0) != 0:
        return False
    if case.kind == AFFIX and not exp["valid"]:
        return got["valid"] == 0
    return all(got[f] == exp[f] for f in FIELDS[case.kind])


# ---- random generators -----------------------------------------------------------------------

def rand_seq(rng, n, n_rate=0.0):
    alpha = "ACGT"
    s = [rng.choice(alpha) for _ in range(n)]
    if n_rate:
        for i in range(n):
            if rng.random() < n_rate:
                s[i] = "N"
    return "".join(s).encode()


def mutate(rng, s: bytes, rate):
    out = bytearray()
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue                                  # deletion
        if x < 2 * rate / 3:
            out.append(rng.choice(b"ACGT"))           # substitution
        elif x < rate:
            out.append(c)
            out.append(rng.choice(b"ACGT"))           # insertion
        else:
            out.append(c)
    return bytes(out)


def pair(rng, n, rate, n_rate=0.0):
    a = rand_seq(rng, n, n_rate)
    return a, mutate(rng, a, rate)


def random_cases(rng, n_per_kind=20, max_len=600):
    cases = []

    def L(lo=1):
        # skew to short lengths but reach max_len
        return max(lo, int(rng.random() ** 2 * max_len))

    for _ in range(n_per_kind):
        a, b = pair(rng, L(), rng.choice([0.0, 0.01, 0.03, 0.1, 0.3]), rng.choice([0, 0, 0.01]))
        cases.append(Case(ALIGN, a, b))
        # GAP: EST window = two exon ends, genomic = exon end + intron + exon start
        e1, e2 = rand_seq(rng, rng.randint(5, 40)), rand_seq(rng, rng.randint(5, 40))
        intron = b"GT" + rand_seq(rng, rng.randint(0, min(200, max_len))) + b"AG"
        est = mutate(rng, e1 + e2, rng.choice([0.0, 0.03, 0.1]))
        cases.append(Case(GAP, est, e1 + intron + e2))
        a, b = pair(rng, L(), rng.choice([0.0, 0.03, 0.2, 0.7]), rng.choice([0, 0.01]))
        cases.append(Case(ED, a, b))
        a, b = pair(rng, L(), rng.choice([0.0, 0.01, 0.03, 0.06, 0.2]))
        n = max(len(a), len(b))
        cases.append(Case(KBAND, a, b, p0=rng.choice([0, 1, 2, int(n * 0.03) + 1, int(n * 0.04) + 1, n])))
        # LCF: long genomic-like s1, short s2 containing a planted common factor
        s1 = rand_seq(rng, rng.randint(1, max_len * 4), rng.choice([0, 0.002]))
        s2 = bytearray(rand_seq(rng, rng.randint(1, 46), rng.choice([0, 0.05])))
        if len(s1) > 8 and len(s2) > 6 and rng.random() < 0.7:
            k = rng.randint(3, min(len(s2), 30, len(s1)))
            p = rng.randint(0, len(s1) - k); q = rng.randint(0, len(s2) - k)
            s2[q:q + k] = s1[p:p + k]
        cases.append(Case(LCF, s1, bytes(s2)))
        # BORDERS: p = gap on the EST, t = genomic region with an intron in the middle
        p = rand_seq(rng, rng.randint(1, min(110, max_len) if rng.random() < 0.7 else max_len))
        cut = rng.randint(0, len(p))
        t = mutate(rng, p[:cut], 0.05) + b"GT" + rand_seq(rng, rng.randint(0, 150)) + b"AG" + \
            mutate(rng, p[cut:], 0.05)
        lo = rng.randint(0, len(p)); hi = rng.randint(lo, len(p))
        if rng.random() < 0.6:
            lo, hi = 0, len(p)
        tail = rng.choice([b"", b"A", b"GT", b"AG"])
        cases.append(Case(BORDERS, p, t, p0=lo, p1=hi, p2=rng.choice([0, 1, 3, len(p) // 10 + 1, len(p)]),
                          b_tail=tail))
        a, b = pair(rng, L(), rng.choice([0.0, 0.03, 0.1, 0.3]))
        if a and b and rng.random() < 0.5:     # the caller only asks when the first chars differ
            b = bytes([a[0] ^ 6]) + b[1:]
        cases.append(Case(AFFIX, a, b))
    return cases


def edge_cases():
    """Empty / single-character / equal / all-N / ragged inputs for every kind."""
    S = [b"", b"A", b"N", b"ACGT", b"AAAAAAAAAA", b"ACGTNNACGT", b"ACGTACGTACGTACGTACGT"]
    cases = []
    for a in S:
        for b in S:
            cases.append(Case(ALIGN, a, b))
            cases.append(Case(GAP, a, b))
            cases.append(Case(ED, a, b))
            for ub in (0, 1, 3, 50):
                cases.append(Case(KBAND, a, b, p0=ub))
            cases.append(Case(LCF, a, b))
            cases.append(Case(AFFIX, a, b))
            for me in (0, 2, 30):
                cases.append(Case(BORDERS, a, b, p0=0, p1=len(a), p2=me))
                cases.append(Case(BORDERS, a, b, p0=len(a) // 2, p1=len(a), p2=me, b_tail=b"GT"))
    # row-class boundaries of the wave kernels: 64*R rows, R = 1,2,4,...
    rng = random.Random(5)
    for n in (63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1024, 1025):
        a, b = pair(rng, n, 0.03)
        cases.append(Case(ALIGN, a, b))
        cases.append(Case(ED, a, b))
        cases.append(Case(AFFIX, a[: min(n, 300)], b[: min(n, 300)]))
        if n <= 600:
            cases.append(Case(GAP, a, rand_seq(rng, 70) + b))
            cases.append(Case(BORDERS, a[:n], b + rand_seq(rng, 40), p0=0, p1=min(n, len(a)), p2=n // 10))
        cases.append(Case(KBAND, a, b, p0=n // 25 + 1))
    return cases
