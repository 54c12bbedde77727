"""GPU box: do the kernel families of one plan overlap?  Wall time of launch+sync of a mixed plan
against the event times of its groups (sum vs max)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pintron_amd import capi
from tools.microbench_dp import rnd, mutate, KIND

rng = random.Random(7)
gen = rnd(200000, rng)
with capi.Context(0) as ctx:
    idx = capi.Index(ctx, gen)
    def plan_of(fams):
        jl = capi.JobList()
        for fam in fams:
            for _ in range(24):
                if fam == "AFFIX":
                    a = rnd(300, rng); jl.add(KIND[fam], a, mutate(a, 0.03, rng))
                elif fam == "BORDERS":
                    a = rnd(300, rng); jl.add(KIND[fam], a, mutate(a, 0.03, rng) + rnd(15000, rng), p0=1, p1=299, p2=15, b_tail=b"AC")
                elif fam == "ALIGN":
                    a = rnd(250, rng); jl.add(KIND[fam], a, mutate(a, 0.03, rng))
                elif fam == "LCF":
                    jl.add(KIND[fam], gen[:120000], rnd(46, rng), a_gen_off=0)
        return jl
    for fams in (["AFFIX"], ["BORDERS"], ["ALIGN"], ["LCF"], ["AFFIX", "BORDERS"], ["AFFIX", "BORDERS", "ALIGN", "LCF"]):
        jl = plan_of(fams)
        best = None
        for rep in range(6):
            p = capi.Plan(ctx, jl, idx)
            t0 = time.perf_counter(); p.launch(); t1 = time.perf_counter(); p.sync(); t2 = time.perf_counter()
            g = {x["name"]: round(x["ms"] * 1000) for x in p.groups() if x["jobs"]}
            p.fetch(); p.close()
            if rep and (best is None or t2 - t0 < best[0]): best = (t2 - t0, t1 - t0, g)
        print("%-32s wall %6.0f us (launch call %4.0f us)  groups(us) %s" % ("+".join(fams), best[0] * 1e6, best[1] * 1e6, best[2]), flush=True)
