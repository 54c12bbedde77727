#!/usr/bin/env python3
"""Per-batch cost of the DP plan API on a batch shaped like one scheduler round (GPU box)."""
import ctypes as C, json, gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pintron_amd.capi as capi
import golden_cases as G
pairs = G.load()
ctx = capi.Context(0)
ctx.L.pgpu_set_timing(ctx.h, 0)
import random
rng = random.Random(1)
for nj in (64, 256, 1024, 3000):
    cases = [pairs[rng.randrange(len(pairs))][0] for _ in range(nj)]
    jl = capi.JobList()
    for c in cases: c.add_to(jl)
    jobs, arena = jl.arrays()
    res = (capi.DpResult * nj)()
    T = dict(create=0, launch=0, sync=0, fetch=0, destroy=0)
    reps = 30
    sbuf = C.create_string_buffer(1 << 24)
    for r in range(reps + 3):
        h = C.c_void_p()
        t0 = time.perf_counter(); ctx.check(ctx.L.pgpu_dp_plan_create(ctx.h, None, jobs, nj, arena, len(arena), C.byref(h)))
        t1 = time.perf_counter(); ctx.check(ctx.L.pgpu_dp_plan_launch(ctx.h, h))
        t2 = time.perf_counter(); ctx.check(ctx.L.pgpu_dp_plan_sync(ctx.h, h))
        t3 = time.perf_counter(); ctx.check(ctx.L.pgpu_dp_plan_fetch(ctx.h, h, res, sbuf, len(sbuf)))
        t4 = time.perf_counter(); ctx.L.pgpu_dp_plan_destroy(ctx.h, h)
        t5 = time.perf_counter()
        if r >= 3:
            T["create"] += t1 - t0; T["launch"] += t2 - t1; T["sync"] += t3 - t2; T["fetch"] += t4 - t3; T["destroy"] += t5 - t4
    print(nj, {k: round(v / reps * 1e3, 3) for k, v in T.items()}, "ms; total", round(sum(T.values()) / reps * 1e3, 3))
