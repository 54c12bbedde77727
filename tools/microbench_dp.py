"""GPU box: latency of one DP launch per family, row count and job count (event time of the group).

Usage: python tools/microbench_dp.py [reps]
Prints one line per case: family, rows x columns, jobs, group name, microseconds (median of reps), and
ns per sweep step (rows-on-lanes sweeps take about rows/R + columns steps)."""
import os
import random
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pintron_amd import capi  # noqa: E402

KIND = dict(ALIGN=0, GAP=1, ED=2, KBAND=3, LCF=4, BORDERS=5, AFFIX=6)


def rnd(n, rng):
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def mutate(s, rate, rng):
    out = bytearray()
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(b"ACGT"))
        if x < rate:
            out.append(rng.choice(b"ACGT"))
            continue
        out.append(c)
    return bytes(out)


def case(ctx, fam, la, lb, njobs, reps, rng):
    jl = capi.JobList()
    for _ in range(njobs):
        a = rnd(la, rng)
        b = mutate(a, 0.03, rng)[:lb] if lb <= la + 8 else mutate(a, 0.03, rng) + rnd(lb - la, rng)
        if fam == "BORDERS":
            jl.add(KIND[fam], a, b, p0=1, p1=la - 1, p2=max(2, la // 20), b_tail=b"AC")
        elif fam == "KBAND":
            jl.add(KIND[fam], a, b, p0=8)
        elif fam == "GAP":
            jl.add(KIND[fam], a, b)
        else:
            jl.add(KIND[fam], a, b)
    times = {}
    for _ in range(reps):
        p = capi.Plan(ctx, jl)
        p.launch(); p.sync()
        for g in p.groups():
            if g["jobs"]:
                times.setdefault(g["name"], []).append(g["ms"] * 1000.0)
        p.fetch()
        p.close()
    for name, t in times.items():
        us = statistics.median(t[1:] or t)
        print("%-8s %5d x %-6d jobs %4d  %-14s %8.1f us   %6.1f ns/(row+col)" %
              (fam, la, lb, njobs, name, us, 1000.0 * us / (la + lb)), flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    rng = random.Random(5)
    os.environ.setdefault("PGPU_MERGED", "0")
    with capi.Context(0) as ctx:
        for fam, shapes in (
            ("AFFIX", [(23, 23), (60, 60), (96, 95), (200, 200), (400, 400), (900, 900)]),
            ("BORDERS", [(17, 15000), (60, 15000), (105, 15000), (260, 15000), (470, 15000), (900, 20000)]),
            ("ALIGN", [(60, 60), (100, 100), (250, 250), (600, 600)]),
            ("GAP", [(60, 200), (120, 400)]),
            ("ED", [(20, 20), (100, 100), (330, 330)]),
            ("KBAND", [(120, 120), (250, 250)]),
        ):
            for la, lb in shapes:
                for nj in (1, 32, 256):
                    case(ctx, fam, la, lb, nj, reps, rng)


if __name__ == "__main__":
    main()
