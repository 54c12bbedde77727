#!/usr/bin/env python3
"""Parity stress on the GPU box: the same inputs through pintron_amd/bin/est-fact many times, across the
library's launch modes, worker counts and with the MEG stage on/off; every run's five files must have the
checksums of the reference object code's run (oracle/_ref/est-fact-core).

On a mismatch everything needed to bisect it is kept under --out: the differing files of both sides, a
unified diff, the environment, the PINTRON_VERBOSE stderr, and -- for the runs made with PINTRON_DP_TRACE
(every second run) -- the verdict of tools/replay_dp_trace.py: did a DP answer differ from the oracle
(kernel / library) or not (host logic / scheduler)?

TEST INFRASTRUCTURE (uses oracle/).  python tools/stress_parity.py --long 30 --c3 10 --out gpurun_out/stress"""
import argparse
import difflib
import hashlib
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.environ.get("PINTRON_STRESS_EXE") or os.path.join(ROOT, "pintron_amd", "bin", "est-fact")
REF = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
FILES = ["raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt"]

MATRIX = [dict(PGPU_MERGED=m, PINTRON_THREADS=t, PINTRON_GPU_MEG=g)
          for m, t, g in itertools.product(("2", "1", "0"), ("18", "2"), ("1", "0"))]
for _k, _e in enumerate(MATRIX):                 # every third setting without the end-exon alignments' exon checks
    if _k % 3 == 2:
        _e["PINTRON_ENDPOINT_CHECKS"] = "0"


def md5s(d):
    return {f: hashlib.md5(open(os.path.join(d, f), "rb").read()).hexdigest() for f in FILES}


def make_inputs(kind, d, n_c3):
    from pintron_amd import synth
    os.makedirs(d, exist_ok=True)
    if kind == "long":
        g, e = synth.make_long_transcripts()
        open(os.path.join(d, "genomic.txt"), "w").write(g)
        open(os.path.join(d, "ests.txt"), "w").write(e)
    else:
        synth.write_files(synth.make("C3", n_est=n_c3), d)


def keep_failure(out, tag, run_dir, ref_dir, env, stderr, want, got, trace):
    dst = os.path.join(out, tag)
    os.makedirs(dst, exist_ok=True)
    report = dict(env=env, differing=[f for f in FILES if want[f] != got.get(f)])
    for f in report["differing"]:
        mine, ref = os.path.join(run_dir, f), os.path.join(ref_dir, f)
        if os.path.exists(mine):
            shutil.copy(mine, os.path.join(dst, "mine-" + f))
        shutil.copy(ref, os.path.join(dst, "ref-" + f))
        if os.path.exists(mine):
            a = open(ref, errors="replace").read().splitlines()
            b = open(mine, errors="replace").read().splitlines()
            diff = list(itertools.islice(difflib.unified_diff(a, b, "ref/" + f, "mine/" + f, lineterm="", n=4), 400))
            open(os.path.join(dst, f + ".diff"), "w").write("\n".join(x[:600] for x in diff) + "\n")
    open(os.path.join(dst, "stderr.txt"), "w").write(stderr)
    if trace and os.path.exists(trace):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_dp_trace.py"), trace, "--json",
                            os.path.join(dst, "dp_replay.json")], capture_output=True, text=True)
        report["dp_replay"] = r.stdout.strip().splitlines()[:30]
        if os.path.getsize(trace) < 48 << 20:
            shutil.copy(trace, os.path.join(dst, "dp_trace.bin"))
    json.dump(report, open(os.path.join(dst, "report.json"), "w"), indent=1)
    return report


POISON = []


def stress(kind, runs_per_env, out, n_c3, matrix, log):
    work = tempfile.mkdtemp(prefix="pintron_stress_%s_" % kind)
    ref_dir, run_dir = os.path.join(work, "ref"), os.path.join(work, "run")
    make_inputs(kind, ref_dir, n_c3)
    os.makedirs(run_dir)
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(ref_dir, f), run_dir)
    subprocess.run([REF], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    want = md5s(ref_dir)
    n_runs = n_bad = 0
    failures = []
    t0 = time.time()
    for ei, env in enumerate(matrix):
        for k in range(runs_per_env):
            for f in FILES:
                p = os.path.join(run_dir, f)
                if os.path.exists(p):
                    os.remove(p)
            e = dict(os.environ, PINTRON_VERBOSE="1", **env)
            if POISON:
                e["PGPU_POISON"] = POISON[n_runs % len(POISON)]
                env = dict(env, PGPU_POISON=e["PGPU_POISON"])
            trace = None
            if k % 2 == 1:
                trace = os.path.join(work, "trace.bin")
                e["PINTRON_DP_TRACE"] = trace
            r = subprocess.run([EXE], cwd=run_dir, env=e, capture_output=True, text=True)
            n_runs += 1
            got = md5s(run_dir) if r.returncode == 0 and all(os.path.exists(os.path.join(run_dir, f)) for f in FILES) else {}
            if got != want:
                n_bad += 1
                tag = "%s_env%02d_run%03d" % (kind, ei, k)
                rep = keep_failure(out, tag, run_dir, ref_dir, env, "rc=%d\n" % r.returncode + r.stderr[-20000:], want, got, trace)
                failures.append(dict(tag=tag, **rep))
                log("MISMATCH %s: %s" % (tag, json.dumps(rep)[:2000]))
        log("%s env %d/%d %s: %d runs so far, %d bad, %.0f s" % (kind, ei + 1, len(matrix), env, n_runs, n_bad, time.time() - t0))
    shutil.rmtree(work, ignore_errors=True)
    return dict(input=kind, runs=n_runs, mismatches=n_bad, failures=failures, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", type=int, default=30, help="runs of the long-transcript input per environment")
    ap.add_argument("--c3", type=int, default=10, help="runs of the C3 sample per environment")
    ap.add_argument("--c3-ests", type=int, default=2000)
    ap.add_argument("--envs", type=int, default=len(MATRIX), help="use the first N environments of the matrix")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress"))
    ap.add_argument("--poison", default=None, help="comma-separated PGPU_POISON bytes cycled over the runs (e.g. 255,165,1): the "
                    "DP plans' strings and traceback workspace start from that pattern -- an answer that depends on it is a read of "
                    "memory nothing wrote")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if args.poison:
        POISON.extend(args.poison.split(","))
    logf = open(os.path.join(args.out, "stress.log"), "a")

    def log(msg):
        print(msg, flush=True)
        logf.write(msg + "\n"); logf.flush()

    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/est-fact-core is not built")
    res = []
    if args.long > 0:
        res.append(stress("long", args.long, args.out, args.c3_ests, MATRIX[:args.envs], log))
    if args.c3 > 0:
        res.append(stress("c3", args.c3, args.out, args.c3_ests, MATRIX[:args.envs], log))
    summary = dict(results=res, matrix=MATRIX[:args.envs])
    json.dump(summary, open(os.path.join(args.out, "summary.json"), "w"), indent=1)
    log("SUMMARY " + json.dumps([{k: v for k, v in r.items() if k != "failures"} for r in res]))
    return 1 if any(r["mismatches"] for r in res) else 0


if __name__ == "__main__":
    sys.exit(main())
