#!/usr/bin/env python3
"""CPU only: the host program under AddressSanitizer + UndefinedBehaviorSanitizer.

Two instrumented builds (outputs under gpurun_out/scratch/asan/): the sequential check program (host
logic + oracle backend) and the product's own main + fibre scheduler + GPU backend code over the CPU
stand-in of the C-ABI (tests/hostcheck/fake_pgpu.c), 3 workers x 16 fibres, clean exit, records
file.  Inputs: test-AMBN, the two sets of real ESTs, C2/C3 samples, the edge-case / long-transcript /
region-start generators.  Prints one line per run; any sanitizer report, a non-zero exit or a
difference between the two builds' raw-multifasta-out.txt is an error."""
import gzip
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pintron_amd import synth  # noqa: E402

H, O, T = (os.path.join(ROOT, p) for p in ("pintron_amd/host", "oracle", "tests/hostcheck"))
OUT = os.path.join(ROOT, "gpurun_out", "scratch", "asan")
shutil.rmtree(OUT, ignore_errors=True)
os.makedirs(OUT)
host = [os.path.join(H, f) for f in ("ef_io.c", "ef_meg.c", "ef_config.c", "ef_fact.c", "ef_refine_intron.c",
                                      "ef_factref.c", "ef_classify.c", "ef_estfact.c", "ef_records.c")]
orc = [os.path.join(O, f) for f in ("pairing_oracle.c", "dp_oracle.c", "dp_oracle_batch.c")]
flags = ["gcc", "-std=gnu99", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread"]
seq = os.path.join(OUT, "estfact_check_asan")
sched = os.path.join(OUT, "estfact_sched_asan")
subprocess.run(flags + ["-o", seq, os.path.join(T, "estfact_check_main.c")] + host + orc + ["-lm"], check=True, cwd=T)
subprocess.run(flags + ["-o", sched, os.path.join(H, "est_fact_main.c"), os.path.join(H, "ef_multi.c"), os.path.join(H, "ef_gpu_backend.c"),
                        os.path.join(H, "ef_sched.c")] + host + [os.path.join(T, "fake_pgpu.c")] + orc + ["-lm"],
               check=True, cwd=T)

cases = {"edge": synth.make_edge_cases(), "long": synth.make_long_transcripts(), "t0": synth.make_region_start_repeats()}
for cfg, n in (("C2", 300), ("C3", 400)):
    w = synth.make(cfg, n_est=n)
    cases[cfg.lower()] = (w.genomic_fasta(), w.ests_fasta())
gold = os.path.join(ROOT, "tests", "golden")
cases["ambn"] = (open(gold + "/ambn/genomic.txt").read(), open(gold + "/ambn/ests.txt").read())
for sub in ("issue13", "example"):
    cases[sub] = tuple(gzip.open("%s/%s/%s.gz" % (gold, sub, f)).read().decode("latin-1") for f in ("genomic.txt", "ests.txt"))

env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0", UBSAN_OPTIONS="print_stacktrace=1",
           PINTRON_THREADS="3", PINTRON_FIBERS="16", PINTRON_CLEAN_EXIT="1", PINTRON_RECORDS_FILE="records.bin")
failed = 0
for name, (g, e) in cases.items():
    outs = {}
    for tag, exe in (("seq", seq), ("sched", sched)):
        d = os.path.join(OUT, name + "_" + tag)
        os.makedirs(d)
        open(d + "/genomic.txt", "w", encoding="latin-1").write(g)
        open(d + "/ests.txt", "w", encoding="latin-1").write(e)
        r = subprocess.run([exe], cwd=d, env=env, capture_output=True, text=True, errors="replace")
        bad = [ln for ln in r.stderr.splitlines() if "runtime error" in ln or "AddressSanitizer" in ln]
        path = d + "/raw-multifasta-out.txt"
        outs[tag] = open(path, "rb").read() if os.path.exists(path) else None
        print("%-8s %-5s rc %d, sanitizer reports %d" % (name, tag, r.returncode, len(bad)))
        for ln in bad[:5]:
            print("    " + ln[:200])
        failed += r.returncode != 0 or bool(bad)
    failed += outs["seq"] != outs["sched"]
sys.exit(1 if failed else 0)
