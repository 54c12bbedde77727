"""GPU box: host time of pgpu_dp_plan_create / launch / sync+fetch for a batch shaped like the bench's
(the first N jobs of the reference's C3 sample calls, tests/golden/c3_sample_jobs.jsonl.gz)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pintron_amd import capi, synth
import golden_cases as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2700
pairs = [p for p in G.load_c3_sample() if p[0].kind != 4][:N]        # (LCF needs the index: left out)
jl = capi.JobList()
for c, _ in pairs:
    c.add_to(jl)
jobs, arena = jl.arrays()
with capi.Context(0) as ctx:
    L = ctx.L
    t = {"create": 0.0, "launch": 0.0, "sync": 0.0, "fetch": 0.0}
    reps = 60
    for r in range(reps + 5):
        h = C.c_void_p()
        t0 = time.perf_counter()
        ctx.check(L.pgpu_dp_plan_create(ctx.h, None, jobs, len(jl.jobs), arena, len(arena), C.byref(h)))
        t1 = time.perf_counter()
        ctx.check(L.pgpu_dp_plan_launch(ctx.h, h))
        t2 = time.perf_counter()
        ctx.check(L.pgpu_dp_plan_sync(ctx.h, h))
        t3 = time.perf_counter()
        nbytes = L.pgpu_dp_plan_string_bytes(h)
        res = (capi.DpResult * len(jl.jobs))(); sbuf = C.create_string_buffer(max(nbytes, 1))
        t3b = time.perf_counter()
        ctx.check(L.pgpu_dp_plan_fetch(ctx.h, h, res, sbuf, nbytes))
        t4 = time.perf_counter()
        L.pgpu_dp_plan_destroy(ctx.h, h)
        if r >= 5:
            t["create"] += t1 - t0; t["launch"] += t2 - t1; t["sync"] += t3 - t2; t["fetch"] += t4 - t3b
    print("%d jobs, arena %d B: " % (len(jl.jobs), len(arena)) + ", ".join("%s %.0f us" % (k, 1e6 * v / reps) for k, v in t.items()))
