#!/usr/bin/env python3
"""bench.py -- est-fact hot path on MI355X: one step = one pass of the accelerated stages over one
C3 batch (BASELINE.json configs[2]: 200 kb genomic x 100 000 ESTs ~600 bp, 3 % errors).

What a step runs TODAY (round 1), all through the C-ABI (libpintron_gpu.so), inputs resident in HBM:
  1. the pairing stage (build_vertex_set replacement) for the 100 000 ESTs of the batch over the
     device suffix-array index of the 200 kb genomic;
  2. the batched DP stage -- every dynamic-programming call the reference est-fact makes for the
     batch (ALIGN, GAP, ED/EDM, KBAND, BORDERS, LCF; AFFIX when present).  The job mix is the
reference's own: tests/golden/c3_sample_jobs.jsonl.gz holds the DP calls of the unmodified
reference on a seeded 400-EST C3 sample (tools/make_bench_fixture.py), tiled to 100 000 ESTs.
The MEG/embedding/filter host logic of est-fact is NOT in the step yet; `config.stages` says so
and the value must be read as the throughput of the accelerated stages, not of the whole program.

Contract: python bench.py --gpus N --steps K --warmup W ; one JSON line on rank 0.
"""
import argparse
import ctypes as C
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

JOB_DT = np.dtype([("kind", "<u4"), ("flags", "<u4"), ("a_off", "<u8"), ("b_off", "<u8"),
                   ("a_len", "<u4"), ("b_len", "<u4"), ("p0", "<u4"), ("p1", "<u4"),
                   ("p2", "<u4"), ("tail", "<u4")], align=True)
assert JOB_DT.itemsize == 48
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_EST_BATCH = 100_000          # C3
REC_BYTES_PER_EST = 256        # size of the per-EST record block gathered to rank 0 (N > 1)


def load_tile(genomic: bytes):
    """One tile = the reference's DP calls for the fixture's ESTs, as (jobs, arena, n_est).
    Genomic-side operands that are exact slices of the genomic are addressed in the resident
    genomic (PGPU_JOB_*_GENOMIC) as the host program does; EST-side strings go to the arena."""
    import pintron_amd.capi as capi
    kinds = dict(ALIGN=capi.ALIGN, GAP=capi.GAP, ED=capi.ED, EDM=capi.ED, KBAND=capi.KBAND,
                 BORDERS=capi.BORDERS, LCF=capi.LCF)
    rows, chunks, size, meta = [], [], 0, None
    with gzip.open(os.path.join(ROOT, "tests", "golden", "c3_sample_jobs.jsonl.gz"), "rt") as f:
        for line in f:
            r = json.loads(line)
            if r["k"] == "META":
                meta = r
                continue
            k = kinds[r["k"]]
            flags = 0
            b = r["b"].encode("latin1")
            p0 = p1 = p2 = tail = 0
            if k == capi.LCF:
                a_off, a_len, flags = 0, r["a_gen_len"], capi.JOB_A_GENOMIC
            else:
                a = r["a"].encode("latin1")
                a_off, a_len = size, len(a)
                chunks.append(a)
                size += len(a)
            if k == capi.KBAND:
                p0 = r["ub"]
            b_tail = b""
            if k == capi.BORDERS:
                p0, p1, p2 = r["min_cut"], r["max_cut"], r["max_errs"]
                b_tail = r["b_tail"].encode("latin1")
                tail = len(b_tail)
            g = genomic.find(b + b_tail) if len(b) >= 24 else -1
            if g >= 0:
                b_off, flags = g, flags | capi.JOB_B_GENOMIC
            else:
                b_off = size
                chunks.append(b + b_tail)
                size += len(b) + len(b_tail)
            rows.append((k, flags, a_off, b_off, a_len, len(b), p0, p1, p2, tail))
    return np.array(rows, dtype=JOB_DT), b"".join(chunks), meta


def tile_jobs(jobs, arena, n_tiles):
    import pintron_amd.capi as capi
    out = np.tile(jobs, n_tiles)
    t = np.repeat(np.arange(n_tiles, dtype=np.uint64), len(jobs)) * np.uint64(len(arena))
    out["a_off"] += np.where(out["flags"] & capi.JOB_A_GENOMIC, np.uint64(0), t)
    out["b_off"] += np.where(out["flags"] & capi.JOB_B_GENOMIC, np.uint64(0), t)
    return out, arena * n_tiles


def make_plan(ctx, idx, jobs, arena):
    import pintron_amd.capi as capi
    h = C.c_void_p()
    ctx.check(ctx.L.pgpu_dp_plan_create(ctx.h, idx.h, C.cast(jobs.ctypes.data, C.POINTER(capi.DpJob)),
                                        len(jobs), arena, len(arena), C.byref(h)))
    plan = capi.Plan.__new__(capi.Plan)
    plan.ctx, plan.n, plan.h, plan._jobs, plan._arena = ctx, len(jobs), h, jobs, arena
    return plan


def cpu_baseline(jobs, arena, genomic, n_est_tile, passes):
    """The CPU oracle (our port of the reference DPs, single thread) on the SAME jobs: one tile =
    the DP calls of `n_est_tile` ESTs.  Checker code, timed here only as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import pintron_amd.capi as capi
    L = O.oracle()
    L.orc_dp_batch.restype = C.c_uint64
    L.orc_dp_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_void_p,
                               C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    res = (capi.DpResult * len(jobs))()
    cap = int(2 * (jobs["a_len"].astype(np.int64) + jobs["b_len"] + 1)[jobs["kind"] <= 1].sum()) + 16
    sbuf = C.create_string_buffer(cap)
    used = C.c_size_t()
    t0 = time.perf_counter()
    cells = 0
    for _ in range(passes):
        cells = L.orc_dp_batch(jobs.ctypes.data, len(jobs), arena, genomic, res, sbuf, cap, C.byref(used))
    dt = time.perf_counter() - t0
    return dict(ests_per_s=n_est_tile * passes / dt, cells=int(cells), seconds=dt,
                mcells_per_s=cells * passes / dt / 1e6), res, sbuf.raw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ests", type=int, default=N_EST_BATCH, help="ESTs per GPU per step (default: C3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)

    import pintron_amd.capi as capi
    from pintron_amd import synth
    ctx = capi.Context(local if world > 1 else 0)       # raises without the HIP library / a GPU
    wl = synth.make("C3", n_est=args.ests)                # seeded C3: 200 kb genomic + ESTs
    genomic = wl.genomic
    t0 = time.perf_counter()
    idx = capi.Index(ctx, genomic)
    t_index = time.perf_counter() - t0
    # patterns as est-fact sees them after strand handling (src/main-est-fact.c:190-213):
    # /clone_end=5' ESTs are reverse-complemented
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    pats = [s.translate(comp)[::-1] if t["rc"] else s for s, t in zip(wl.est_seqs, wl.truth)]
    pplan = capi.PairingPlan(ctx, idx, pats)
    tile, tile_arena, meta = load_tile(genomic)
    n_tiles = max(1, -(-args.ests // meta["n_est"]))
    n_est = n_tiles * meta["n_est"]
    jobs, arena = tile_jobs(tile, tile_arena, n_tiles)
    t0 = time.perf_counter()
    plan = make_plan(ctx, idx, jobs, arena)
    t_upload = time.perf_counter() - t0

    rec = None
    gather_list = None
    if world > 1:
        rec = torch.empty(n_est * REC_BYTES_PER_EST, dtype=torch.uint8, device="cuda")
        if rank == 0:
            gather_list = [torch.empty_like(rec) for _ in range(world)]

    def step():
        pplan.run(15, 0.2)                 # options.ggo defaults: -l 15, -d 0.2
        plan.launch()
        plan.sync()
        if world > 1:
            # per-EST record block -> rank 0 over RCCL (the only exchange of the sharded path)
            ctx.check(ctx.L.pgpu_dp_plan_results_to_device(ctx.h, plan.h, rec.data_ptr(), rec.numel()))
            dist.gather(rec, gather_list, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    groups_acc = None
    pair_ms = {k: 0.0 for k in capi.PairingPlan.STAGES}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in pplan.stage_ms().items():
            pair_ms[k] += v / args.steps
        g = plan.groups()
        if groups_acc is None:
            groups_acc = g
        else:
            for x, y in zip(groups_acc, g):
                x["ms"] += y["ms"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    for x in groups_acc:
        x["ms"] /= args.steps

    out = None
    if rank == 0:
        # reference-accurate cell count of one step: from the oracle's own accounting on a tile
        base = None
        if world == 1 and not args.no_cpu:
            base, cres, cstr = cpu_baseline(tile, tile_arena, genomic, meta["n_est"], passes=8)
            # the same tile on the GPU must agree with what the CPU just computed (bit-exact)
            res, strings = plan.fetch()
            for i in range(len(tile)):
                k = int(tile["kind"][i])
                a, b = capi.decode(k, res[i], strings), capi.decode(k, cres[i], cstr)
                if a != b:
                    raise SystemExit("bench: GPU result %d differs from the oracle: %r vs %r" % (i, a, b))
        dom = max(groups_acc, key=lambda g: g["ms"])
        tile_cells = base["cells"] if base else None
        ms_step = dt / args.steps * 1e3
        value = n_est * world / (dt / args.steps)
        out = {
            "metric": "ESTs aligned/sec (whole node) + DP Mcells/s; bit-exact factorizations vs ref",
            "value": value, "unit": "ESTs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": "C3: 200 kb genomic x %d ESTs ~600 bp, 3%% errors, per GPU" % n_est,
                       "stages": "pairing stage (%d pattern positions, %d pairings) + batched DP stage (reference's own DP"
                                 " call mix, %d calls) per step per GPU; MEG/embedding/filter host logic not yet in the step"
                                 % (sum(len(x) for x in pats), pplan.ctx.L.pgpu_pairing_plan_count(pplan.h), len(jobs)),
                       "ests_per_gpu": n_est, "dp_jobs_per_gpu": int(len(jobs)), "parallelism": "est-shard x%d" % world},
            "dp_mcells_per_s": (tile_cells * n_tiles * world / (dt / args.steps) / 1e6) if tile_cells else None,
            "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": dom["algo_bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (dom["algo_bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom["ms"] else None,
                         "traffic": None, "avg_launch_ms": dom["ms"], "algo_bytes_per_launch": dom["algo_bytes"]},
            "kernels": [{"name": g["name"], "jobs": g["jobs"], "ms": round(g["ms"], 4),
                         "algo_GBs": round(g["algo_bytes"] / (g["ms"] * 1e-3) / 1e9, 1) if g["ms"] else None}
                        for g in groups_acc] +
                       [{"name": "pair_" + k, "jobs": len(pats), "ms": round(v, 4), "algo_GBs": None}
                        for k, v in pair_ms.items()],
            "index_build_s": t_index,
            "upload_s": t_upload,
        }
        if base:
            out["cpu_baseline"] = {"value": base["ests_per_s"], "unit": "ESTs/s", "cores": 1, "kind": "port",
                                   "sample": "DP stage of %d C3 ESTs (one fixture tile, %d DP calls, %.2f Gcells) x8 passes, %.1f s"
                                             % (meta["n_est"], len(tile), base["cells"] / 1e9, base["seconds"]),
                                   "mcells_per_s": base["mcells_per_s"]}
        print(json.dumps(out), flush=True)
    plan.close()
    pplan.close()
    idx.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
