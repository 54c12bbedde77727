"""GPU parity tests proper: every DP kind through the C-ABI (libpintron_gpu.so, HIP/gfx950)
against the CPU oracle and the committed golden vectors.  Bit-exact (integer / byte work)."""
import random

import pytest

import dp_cases as D
import golden_cases as G

pytestmark = pytest.mark.gpu


def run_and_check(ctx, O, cases, expected=None):
    import pintron_amd.capi as capi
    jl = capi.JobList()
    for c in cases:
        c.add_to(jl)
    out = capi.run_jobs(ctx, jl)
    bad = []
    for i, (c, got) in enumerate(zip(cases, out)):
        exp = expected[i] if expected is not None else c.expected(O)
        if not D.check_case(c, got, O, expected=exp):
            bad.append((c, exp, got))
    assert not bad, "%d/%d jobs differ; first: %r\nexpected %r\ngot      %r" % (
        len(bad), len(cases), *bad[0])


def test_golden_vectors(gpu_ctx, O):
    pairs = G.load()
    run_and_check(gpu_ctx, O, [c for c, _ in pairs], [e for _, e in pairs])


def test_c3_sample_golden_vectors(gpu_ctx, O):
    """10 061 DP calls of the unmodified reference on a C3-shaped sample, with ITS answers."""
    pairs = G.load_c3_sample()
    run_and_check(gpu_ctx, O, [c for c, _ in pairs], [e for _, e in pairs])


def test_plan_from_parts_equals_one_plan(gpu_ctx, O):
    """pgpu_dp_plan_create_parts: a batch assembled from several producers (each with its own operand
    arena, one of them empty) gives, part after part, what plans of their own give."""
    import pintron_amd.capi as capi
    rng = random.Random(44)
    groups = [D.random_cases(rng, n_per_kind=4, max_len=200), [], D.random_cases(rng, n_per_kind=2, max_len=500),
              D.random_cases(rng, n_per_kind=1, max_len=60)]
    lists = []
    for cases in groups:
        jl = capi.JobList()
        for c in cases:
            c.add_to(jl)
        lists.append(jl)
    separate = [o for jl in lists if jl.jobs for o in capi.run_jobs(gpu_ctx, jl)]
    plan = capi.Plan.from_parts(gpu_ctx, lists)
    try:
        plan.launch()
        plan.sync()
        res, strings = plan.fetch()
        kinds = [j.kind for jl in lists for j in jl.jobs]
        merged = [capi.decode(kinds[i], res[i], strings) for i in range(plan.n)]
    finally:
        plan.close()
    assert merged == separate and len(merged) == sum(len(g) for g in groups)


def test_edge_cases(gpu_ctx, O):
    run_and_check(gpu_ctx, O, D.edge_cases())


@pytest.mark.parametrize("seed,max_len", [(1, 64), (2, 300), (3, 700), (4, 1500)])
def test_random_cases(gpu_ctx, O, seed, max_len):
    run_and_check(gpu_ctx, O, D.random_cases(random.Random(seed), n_per_kind=60, max_len=max_len))


def test_large_rows(gpu_ctx, O):
    """Row classes R=32 and R=64 (up to 4096 rows) and the dimension limits."""
    import pintron_amd.capi as capi
    rng = random.Random(9)
    cases = []
    for n in (1100, 2047, 2421, 4096):
        a, b = D.pair(rng, n, 0.03, 0.001)
        cases.append(D.Case(D.ALIGN, a, b))
        cases.append(D.Case(D.ED, a, b))
        cases.append(D.Case(D.KBAND, a, b, p0=n // 25))
    a, b = D.pair(rng, 1500, 0.05)
    cases.append(D.Case(D.AFFIX, a, b))
    cases.append(D.Case(D.BORDERS, a[:1100], b, p0=0, p1=1100, p2=60))
    cases.append(D.Case(D.GAP, a[:1500], b + D.rand_seq(rng, 300)))
    run_and_check(gpu_ctx, O, cases)
    # beyond every limit (GAP: 16 000 per side): per-job ERANGE, the rest of the batch still runs
    jl = capi.JobList()
    big = D.rand_seq(rng, 16001)
    jl.add(capi.GAP, big[:100], big)
    jl.add(capi.ED, b"ACGT", b"ACGA")
    jl.add(capi.GAP, big, big[:100])
    out = capi.run_jobs(gpu_ctx, jl)
    assert out[0]["status"] == capi.PGPU_ERANGE
    assert out[1] == dict(status=0, score=1)
    assert out[2]["status"] == capi.PGPU_ERANGE


def test_slow_kernels_beyond_the_fast_row_limits(gpu_ctx, O):
    """GAP windows of more than 2048 EST characters and BORDERS patterns of more than 4096 used to be
    refused (PGPU_ERANGE, est-fact stopped); the reference computes them, so an anti-diagonal kernel
    over HBM now answers them -- slowly, bit-exactly.  Sizes just past the limits and well beyond,
    ragged shapes, empty and one-character operands through the same launch group."""
    rng = random.Random(44)
    cases = []
    for n, m in ((2049, 2300), (2500, 300), (3000, 3200), (4000, 9000)):
        a, b = D.pair(rng, max(n, m), 0.04, 0.002)
        a = a[:n]
        cut = n // 2
        g = b[:cut] + b"GT" + D.rand_seq(rng, max(0, m - n)) + b"AG" + b[cut:n]
        cases.append(D.Case(D.GAP, a, g[:m] if len(g) > m else g))
    for n, extra, errs in ((4097, 50, 40), (5000, 900, 120), (9000, 3000, 200)):
        a, b = D.pair(rng, n, 0.02, 0.001)
        a = a[:n]
        cut = rng.randint(0, n)
        t = D.mutate(rng, a[:cut], 0.02) + b"GT" + D.rand_seq(rng, extra) + b"AG" + D.mutate(rng, a[cut:], 0.02)
        cases.append(D.Case(D.BORDERS, a, t, p0=0, p1=n, p2=errs))
        cases.append(D.Case(D.BORDERS, a, t, p0=n // 3, p1=2 * n // 3, p2=errs, b_tail=b"GT"))
    run_and_check(gpu_ctx, O, cases)


def test_strips_beyond_4096_rows(gpu_ctx, O):
    """More than 4096 rows: the R = 64 kernels sweep the matrix in strips of 4096 rows (long 3' UTR
    exons of full-length mRNAs).  Sizes around the strip boundaries, square and ragged."""
    rng = random.Random(31)
    cases = []
    for n, m in ((4097, 4097), (5000, 700), (8192, 8300), (8193, 40), (9000, 9100), (12289, 6000)):
        a, b = D.pair(rng, max(n, m), 0.03, 0.001)
        a, b = a[:n], b[:m]
        cases.append(D.Case(D.ALIGN, a, b))
        cases.append(D.Case(D.ED, a, b))
        cases.append(D.Case(D.ED, b, a))
        cases.append(D.Case(D.KBAND, a, b, p0=max(n, m) // 25))
        cases.append(D.Case(D.KBAND, a, b, p0=max(n, m)))
        cases.append(D.Case(D.AFFIX, a, b))
    big = D.rand_seq(rng, 6000)
    cases.append(D.Case(D.ALIGN, big, big))                  # identity shortcut at that size
    cases.append(D.Case(D.ALIGN, big, b""))
    run_and_check(gpu_ctx, O, cases)


def test_cooperative_kernels_at_class_boundaries(gpu_ctx, O):
    """BORDERS / AFFIX with more than 64 rows run one job per workgroup (4 waves per sweep): row
    counts around every lane/wave/row-class boundary, short and long column counts, sub-ranges."""
    rng = random.Random(21)
    cases = []
    for n in (63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 511, 512, 513, 767, 1023, 1024,
              1025, 1500, 2048, 2049, 4095, 4096):
        a, b = D.pair(rng, n, rng.choice([0.0, 0.03, 0.1]), 0.001)
        a = a[:n]
        cases.append(D.Case(D.AFFIX, a, b))
        cases.append(D.Case(D.AFFIX, a, b[: max(1, len(b) // 7)]))
        cut = rng.randint(0, len(a))
        t = D.mutate(rng, a[:cut], 0.04) + b"GT" + D.rand_seq(rng, rng.randint(0, 400)) + b"AG" + D.mutate(rng, a[cut:], 0.04)
        for (lo, hi) in ((0, len(a)), (len(a) // 3, 2 * len(a) // 3), (len(a), len(a)), (5, 5)):
            if lo <= hi <= len(a):
                cases.append(D.Case(D.BORDERS, a, t, p0=lo, p1=hi, p2=rng.choice([0, 3, len(a) // 10 + 1, len(a)]),
                                    b_tail=rng.choice([b"", b"A", b"GT"])))
        cases.append(D.Case(D.BORDERS, a, t[: max(1, len(t) // 5)], p0=0, p1=len(a), p2=2))
    run_and_check(gpu_ctx, O, cases)


def test_lcf_genomic_scale(gpu_ctx, O):
    """LCF at the BASELINE configs' genomic lengths (200 kb and 1 Mb prefixes) with the operand
    taken from the resident genomic; checked against the oracle (seconds on CPU)."""
    import pintron_amd.capi as capi
    rng = random.Random(21)
    gen = D.rand_seq(rng, 1_000_000, 0.0005)
    idx = capi.Index(gpu_ctx, gen)
    try:
        jl = capi.JobList()
        cases = []
        for glen in (199_990, 1_000_000, 65_536, 77):
            for _ in range(3):
                est = bytearray(D.rand_seq(rng, rng.randint(20, 46), 0.02))
                k = rng.randint(8, 18)
                p = rng.randint(0, glen - k); q = rng.randint(0, len(est) - k)
                est[q:q + k] = gen[p:p + k]
                cases.append(D.Case(D.LCF, gen[:glen], bytes(est)))
                jl.add(capi.LCF, gen[:glen], bytes(est), a_gen_off=0)
        out = capi.run_jobs(gpu_ctx, jl, idx)
        for c, got in zip(cases, out):
            assert D.check_case(c, got, O), (len(c.a), c.b, got, c.expected(O))
    finally:
        idx.close()


def test_properties_at_full_size(gpu_ctx, O):
    """Size-independent properties on a C3-shaped batch (20k jobs): ALIGN of a string with
    itself is the identity; alignment strings project back onto their inputs; the alignment's
    mismatch+gap count equals the score; ED is symmetric and equals ALIGN's score when no N is
    present; KBAND agrees with ED whenever ED <= k."""
    import pintron_amd.capi as capi
    rng = random.Random(33)
    jl = capi.JobList()
    meta = []
    for i in range(5000):
        a, b = D.pair(rng, rng.randint(80, 600), 0.03)
        k = max(len(a), len(b)) // 25 + 1
        meta.append((a, b, k))
        jl.add(capi.ALIGN, a, b)
        jl.add(capi.ED, a, b)
        jl.add(capi.ED, b, a)
        jl.add(capi.KBAND, a, b, p0=k)
    out = capi.run_jobs(gpu_ctx, jl)
    for i, (a, b, k) in enumerate(meta):
        al, e1, e2, kb = out[4 * i: 4 * i + 4]
        assert al["ea"].replace(b"-", b"") == a and al["ga"].replace(b"-", b"") == b
        assert len(al["ea"]) == len(al["ga"]) == al["dim"]
        cost = sum(1 for x, y in zip(al["ea"], al["ga"]) if x != y)
        assert cost == al["score"] == e1["score"] == e2["score"]
        assert kb["ok"] == (1 if e1["score"] <= k else 0)
        if kb["ok"]:
            assert kb["edit"] == e1["score"]
    jl = capi.JobList()
    for a, _, _ in meta[:200]:
        jl.add(capi.ALIGN, a, a)
    for o, (a, _, _) in zip(capi.run_jobs(gpu_ctx, jl), meta):
        assert o["score"] == 0 and o["ea"] == a and o["ga"] == a


def test_kband_band_on_lanes(gpu_ctx, O):
    """K-band distances whose band fits a wave (2k+1 <= 64) run with the band on the lanes: every k
    up to 31 (and 32, the first that does not fit), length differences 0..k in both argument orders,
    short and long strings, error rates below and above the bound, wildcard characters (which are
    NOT wildcards here)."""
    rng = random.Random(77)
    cases = []
    for k in (1, 2, 3, 5, 9, 15, 16, 30, 31, 32):
        for n in (2 * k + 2, 2 * k + 3, 3 * k + 7, 64, 65, 129, 300, 700, 1500):
            if 2 * k + 1 >= n:
                continue
            for rate in (0.0, 0.01, 0.04, 0.12):
                a = D.rand_seq(rng, n, rng.choice([0.0, 0.01]))
                b = D.mutate(rng, a, rate)
                d = rng.randint(0, k)                         # force a length difference of up to k
                if len(b) > d and rng.random() < 0.5:
                    b = b[: len(b) - d] if rng.random() < 0.5 else b[d:]
                cases.append(D.Case(D.KBAND, a, b, p0=k))
                cases.append(D.Case(D.KBAND, b, a, p0=k))
    a = D.rand_seq(rng, 400)
    cases.append(D.Case(D.KBAND, a, a[:395] + b"TTTTT", p0=7))
    cases.append(D.Case(D.KBAND, a, a[7:], p0=7))              # the result sits on the last slot of the band
    cases.append(D.Case(D.KBAND, a[7:], a, p0=7))
    cases.append(D.Case(D.KBAND, a, D.rand_seq(rng, 400), p0=20))   # unrelated: far above the bound
    run_and_check(gpu_ctx, O, cases)


def test_exon_check_dust_flags_beside_the_banded_distance(gpu_ctx, O):
    """KBAND jobs with tail = 1 ("exon check", include/pintron_gpu.h): beside K_band_edit_distance's answer the two
    comparisons of clean_low_complexity_exons_2 (src/est-factorizations.c:1687-1691) -- dustScore
    (src/exon-complexity.c:50-78) of each operand, FP64 in the reference's operation order, against a threshold passed
    as its double bits.  Random, low-complexity (homopolymers, microsatellites), lower-case and N-bearing strings,
    lengths 0..3 (score 0) up to thousands (every row class, the strips), thresholds ON the score of one of the
    operands (the comparison is strict) and just beside it; the distance's early exits (equal strings, bound 0,
    length difference above the bound) must not skip the flags."""
    import struct
    import pintron_amd.capi as capi
    rng = random.Random(909)
    jl = capi.JobList()
    want = []

    def lowc(n):
        kind = rng.randrange(4)
        if kind == 0:
            return bytes([rng.choice(b"ACGT")]) * n
        if kind == 1:
            u = D.rand_seq(rng, rng.randint(2, 4))
            return (u * (n // len(u) + 1))[:n]
        if kind == 2:
            return D.rand_seq(rng, n, 0.05).lower()
        return D.rand_seq(rng, n, rng.choice([0.0, 0.02]))

    for it in range(700):
        n = rng.choice([0, 1, 2, 3, 4, 17, 63, 64, 65, 150, 300, 1000, 1100, 2100, 4200, 5000]) if it % 3 == 0 else rng.randint(0, 400)
        g = lowc(n)
        e = D.mutate(rng, g, rng.choice([0.0, 0.0, 0.03, 0.3])) if rng.random() < 0.8 else lowc(rng.randint(0, 400))
        sg, se = O.dust_score(g), O.dust_score(e)
        thr = rng.choice([20.0, 4.5, sg, se, sg * (1 + 1e-15) if sg else 0.25, se * (1 - 1e-15) if se else 0.5, 1e-9, 1e9])
        if thr <= 0.0:
            thr = 0.125
        lo, hi = struct.unpack("<II", struct.pack("<d", thr))
        ub = rng.choice([0, 1, 3, max(len(g), len(e)) // 25 + 1, max(len(g), len(e))])
        jl.add(capi.KBAND, g, e, p0=ub, p1=lo, p2=hi, tail=1)
        exp = O.kband(g, e, ub)
        exp["dust"] = O.dust_flags(g, e, thr)
        want.append((g[:40], e[:40], len(g), len(e), thr, ub, exp))
    out = capi.run_jobs(gpu_ctx, jl)
    bad = [(w, got) for w, got in zip(want, out) if got.get("status", 0) != 0 or any(got[f] != w[6][f] for f in ("ok", "edit", "dust"))]
    assert not bad, (len(bad), bad[0])


def test_align_inside_a_band(gpu_ctx, O):
    """ALIGN above 64 rows first runs inside a band of half-width 31 on one wave (pgpu_dp_kernels.hip:
    align_band_sweep) and falls back to the whole matrix when the banded score exceeds 31: scores on both
    sides of the bound, length differences up to and beyond the band, gaps that push the path to the band's
    edge, ties between diagonal / up / left (low-complexity strings), wildcard characters, strings whose
    lengths sit on the 16-row groups of the direction words."""
    rng = random.Random(4242)
    cases = []
    for n in (65, 66, 79, 80, 81, 95, 96, 97, 128, 200, 250, 255, 256, 257, 300, 511, 512, 513, 600, 1000, 2000):
        for rate in (0.0, 0.005, 0.03, 0.08, 0.12, 0.2, 0.5):
            a = D.rand_seq(rng, n, rng.choice([0.0, 0.0, 0.01]))
            b = D.mutate(rng, a, rate)
            cases.append(D.Case(D.ALIGN, a, b))
            cases.append(D.Case(D.ALIGN, b, a) if len(b) > 64 else D.Case(D.ALIGN, a, b[::-1]))
        a = D.rand_seq(rng, n)
        for d in (1, 15, 30, 31, 32, 33, 60):                     # length differences around the band's half-width
            if n > d + 64:
                cases.append(D.Case(D.ALIGN, a, a[d:]))            # one gap of d at the start
                cases.append(D.Case(D.ALIGN, a[:n - d], a))        # ... at the end
                cases.append(D.Case(D.ALIGN, a[:n // 2] + a[n // 2 + d:], a))   # ... in the middle
                cases.append(D.Case(D.ALIGN, a, a[:n // 3] + D.rand_seq(rng, d) + a[n // 3:]))
        # a deletion and an insertion of the same size far apart: the path leaves the diagonal by g and comes back
        for g in (5, 16, 29, 31, 32, 40):
            if n > 2 * g + 80:
                b = a[:20] + a[20 + g:n - 30] + D.rand_seq(rng, g) + a[n - 30:]
                cases.append(D.Case(D.ALIGN, a, b))
        # low complexity: many co-optimal alignments, the tie order decides
        lc = (b"AC" * n)[:n]
        cases.append(D.Case(D.ALIGN, lc, lc[1:] + b"A"))
        cases.append(D.Case(D.ALIGN, b"A" * n, b"A" * (n - 7)))
        cases.append(D.Case(D.ALIGN, b"A" * (n - 3), b"A" * n))
        cases.append(D.Case(D.ALIGN, D.mutate(rng, lc, 0.05), lc))
        cases.append(D.Case(D.ALIGN, b"N" * n, D.rand_seq(rng, n + 5)))
    run_and_check(gpu_ctx, O, cases)


def test_results_to_device_and_launch_modes(gpu_ctx, O):
    """pgpu_dp_plan_results_to_device hands the COMPLETE table to device memory (the LCF answers are
    decoded on the host from the kernel's keys); and the three launch modes of the library (PGPU_MERGED
    = 2: one batch launch + LCF, 1: wave-per-job launch + sweeps, 0: a launch per family) give the same
    answers for a batch that holds every family, one-job-per-workgroup sweeps included."""
    import ctypes as C
    import os
    import pintron_amd.capi as capi
    rng = random.Random(31)
    cases = D.random_cases(rng, n_per_kind=12, max_len=200)
    for rows in (70, 130, 300, 520):                      # BORDERS / AFFIX above 64 rows: the cooperative sweeps
        a = bytes(rng.choice(b"ACGT") for _ in range(rows))
        b = bytes(c if rng.random() > 0.05 else rng.choice(b"ACGT") for c in a)
        cases.append(D.Case(D.AFFIX, a, b))
        cases.append(D.Case(D.BORDERS, a, b + bytes(rng.choice(b"ACGT") for _ in range(500)), p0=1, p1=rows - 1, p2=rows // 10, b_tail=b"GT"))
    jl = capi.JobList()
    for c in cases:
        c.add_to(jl)
    p = capi.Plan(gpu_ctx, jl)
    p.launch()
    # device memory from the HIP runtime the library itself runs on: whichever copy of it the process mapped first
    # (torch brings its own; a second copy loaded by file name finds no device).  dlopen by SONAME returns the
    # copy that is already loaded, a path from /proc/self/maps is the fallback.
    try:
        hip = C.CDLL("libamdhip64.so.7")
    except OSError:
        with open("/proc/self/maps") as f:
            mapped = sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
        assert mapped, "libpintron_gpu.so is loaded but no libamdhip64 is mapped"
        hip = C.CDLL(mapped[0])
    nbytes = len(cases) * C.sizeof(capi.DpResult)
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), C.c_size_t(nbytes)) == 0
    p.results_to_device(dev.value, nbytes)
    res, strs = p.fetch()
    p.close()
    back = C.create_string_buffer(nbytes)
    assert hip.hipMemcpy(back, dev, C.c_size_t(nbytes), 2) == 0          # hipMemcpyDeviceToHost
    hip.hipFree(dev)
    assert back.raw == bytes(res)[:nbytes]
    base = [capi.decode(c.kind, r, strs) for c, r in zip(cases, res)]
    for c, got in zip(cases, base):
        assert D.check_case(c, got, O), (c, got)
    for mode in ("1", "0"):
        os.environ["PGPU_MERGED"] = mode
        try:
            with capi.Context(0) as ctx2:
                out = capi.run_jobs(ctx2, jl)
        finally:
            del os.environ["PGPU_MERGED"]
        assert out == base, "PGPU_MERGED=%s differs" % mode


@pytest.mark.parametrize("merged", ["2", "1", "0"])
def test_lcf_from_the_suffix_array(O, merged, monkeypatch):
    """find_longest_common_factor_dp of a genomic PREFIX against at most 64 EST characters, answered from
    the resident suffix array (lcfsa_wave_body) instead of the 46 x |prefix| matrix: same length, same first
    maximum (smallest start in the genomic, then in the EST) as the oracle on random sequences, planted
    factors, repeats (microsatellites, duplicated blocks: many suffixes share the pattern, so the range
    minima decide), tiny alphabets, prefixes that cut an occurrence, empty operands; an EST prefix with ONE N
    (upper or lower case) is answered there too -- the four strings with A, C, G, T in its place -- and the jobs
    that do not qualify (two Ns, an N in the genomic prefix, more than 64 characters, a prefix beyond the first
    non-ACGT character) still get the matrix kernel's answer.  In every launch mode of the library."""
    import pintron_amd.capi as capi
    monkeypatch.setenv("PGPU_MERGED", merged)
    rng = random.Random(77 + int(merged))

    def genome(kind, n):
        if kind == "random":
            return D.rand_seq(rng, n)
        if kind == "binary":
            return bytes(rng.choice(b"AC") for _ in range(n))
        if kind == "repeats":
            g = bytearray(D.rand_seq(rng, n))
            unit = D.rand_seq(rng, rng.randint(2, 5))
            for _ in range(max(1, n // 4000)):
                p = rng.randint(0, max(0, n - 400)); k = rng.randint(30, 300)
                g[p:p + k] = (unit * (k // len(unit) + 1))[:k]
            blk = bytes(g[n // 3:n // 3 + 200])
            for _ in range(6):
                p = rng.randint(0, max(0, n - 200)); g[p:p + len(blk)] = blk[: n - p]
            return bytes(g[:n])
        raise ValueError(kind)

    with capi.Context(0) as ctx:
        for kind, n in (("random", 200_000), ("repeats", 60_000), ("binary", 3_000), ("random", 9), ("random", 1)):
            gen = genome(kind, n)
            mixed = gen if n < 50_000 else gen[: n - 5000] + b"N" + gen[n - 4999:]      # first non-ACGT at n - 5000
            idx = capi.Index(ctx, mixed)
            try:
                jl = capi.JobList()
                cases = []
                for it in range(260 if n >= 3000 else 60):
                    G = rng.choice([0, 1, 7, 8, 9, n, n // 2, rng.randint(0, n), rng.randint(0, n)])
                    if it % 9 == 0 and n >= 50_000:
                        G = rng.randint(n - 5000, n)                     # beyond the N: the matrix kernel
                    l2 = rng.choice([0, 1, 5, 8, 9, 23, 46, 46, 46, 64, 65, rng.randint(0, 64)])
                    s2 = bytearray(D.rand_seq(rng, l2, 0.02 if it % 7 == 0 else 0.0))
                    if l2 >= 4 and G >= 4 and rng.random() < 0.7:       # plant a factor of the prefix (sometimes cut by G)
                        k = rng.randint(2, min(l2, 40, G))
                        p = rng.randint(max(0, G - 60), G - 1) if rng.random() < 0.3 else rng.randint(0, G - 1)
                        k = min(k, n - p)
                        q = rng.randint(0, l2 - k)
                        s2[q:q + k] = mixed[p:p + k]
                    # ONE wildcard in the EST prefix (the four substitutions are searched by the same wave): inside the
                    # planted factor, at either end, lower case; now and then a second one (the matrix kernel's case)
                    if len(s2) >= 1 and it % 3 == 1:
                        s2 = bytearray(bytes(s2).replace(b"N", b"A"))
                        s2[rng.choice([0, len(s2) - 1, rng.randrange(len(s2))])] = ord("n") if it % 6 == 1 else ord("N")
                        if it % 15 == 1 and len(s2) >= 2:
                            s2[rng.randrange(len(s2))] = ord("N")
                    cases.append(D.Case(D.LCF, mixed[:G], bytes(s2)))
                    jl.add(capi.LCF, mixed[:G], bytes(s2), a_gen_off=0)
                out = capi.run_jobs(ctx, jl, idx)
                bad = [(len(c.a), c.b, got, c.expected(O)) for c, got in zip(cases, out) if not D.check_case(c, got, O)]
                assert not bad, (kind, n, len(bad), bad[0])
            finally:
                idx.close()
