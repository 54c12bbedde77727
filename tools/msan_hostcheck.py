#!/usr/bin/env python3
"""CPU only: the host program under MemorySanitizer (reads of uninitialised memory: the kind of bug
whose effect depends on which fibre stack or heap block an EST happens to get, i.e. on thread timing).

The product's own main + fibre scheduler + GPU backend code over the CPU stand-in of the C-ABI
(tests/hostcheck/fake_pgpu.c), built with ROCm's clang -fsanitize=memory -fsanitize-memory-track-origins=2
(outputs under gpurun_out/scratch/msan/).  Inputs: test-AMBN, C2/C3 samples, the edge-case / long-transcript /
region-start generators; batched, direct, host MEG, device-MEG-unavailable.  Any report or non-zero exit is
an error."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pintron_amd import synth  # noqa: E402

H, O, T = (os.path.join(ROOT, p) for p in ("pintron_amd/host", "oracle", "tests/hostcheck"))
OUT = os.path.join(ROOT, "gpurun_out", "scratch", "msan")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
shutil.rmtree(OUT, ignore_errors=True)
os.makedirs(OUT)
src = [os.path.join(H, f) for f in sorted(os.listdir(H)) if f.endswith(".c")] + \
      [os.path.join(T, "fake_pgpu.c")] + [os.path.join(O, f) for f in ("pairing_oracle.c", "dp_oracle.c", "dp_oracle_batch.c")]
exe = os.path.join(OUT, "estfact_sched_msan")
subprocess.run([CLANG, "-std=gnu99", "-O1", "-g", "-fsanitize=memory", "-fsanitize-memory-track-origins=2", "-fno-omit-frame-pointer",
                "-pthread", "-o", exe] + src + ["-lm"], check=True)

cases = {"long": synth.make_long_transcripts(), "edge": synth.make_edge_cases(), "t0": synth.make_region_start_repeats(),
         "t0copies": synth.make_region_start_copies()}
for cfg, n in (("C2", 300), ("C3", 600), ("C5", 800)):
    w = synth.make(cfg, n_est=n)
    cases[cfg.lower()] = (w.genomic_fasta(), w.ests_fasta())
gold = os.path.join(ROOT, "tests", "golden", "ambn")
cases["ambn"] = (open(gold + "/genomic.txt").read(), open(gold + "/ests.txt").read())
failed = 0
for name, (g, e) in cases.items():
    for mode in ({}, {"PINTRON_ESTFACT_MODE": "direct"}, {"PINTRON_GPU_MEG": "0"}, {"PINTRON_FAKE_MEG_LIMIT": "6"}):
        d = os.path.join(OUT, "%s_%s" % (name, "_".join(mode.values()) or "batched"))
        os.makedirs(d)
        open(d + "/genomic.txt", "w").write(g)
        open(d + "/ests.txt", "w").write(e)
        r = subprocess.run([exe], cwd=d, env=dict(os.environ, PINTRON_THREADS="3", PINTRON_FIBERS="6", PINTRON_CLEAN_EXIT="1", **mode),
                           capture_output=True, text=True, errors="replace")
        n_rep = sum("MemorySanitizer" in ln for ln in r.stderr.splitlines())
        print("%-9s %-32s rc %d, reports %d" % (name, mode or "batched", r.returncode, n_rep))
        if r.returncode or n_rep:
            failed += 1
            print("\n".join("    " + ln[:200] for ln in r.stderr.splitlines() if "Sanitizer" in ln or " #" in ln)[:3000])
print("MSan: %d failing runs" % failed)
sys.exit(1 if failed else 0)
