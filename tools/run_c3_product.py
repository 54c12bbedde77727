#!/usr/bin/env python3
"""Times the product est-fact binary on a C3-shaped workload (GPU box)."""
import os, subprocess, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pintron_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t = time.time(); w = synth.make("C3", n_est=n); print("synth %.1fs" % (time.time() - t), flush=True)
d = tempfile.mkdtemp(prefix="c3run_")
synth.write_files(w, d)
env = dict(os.environ, PINTRON_VERBOSE="1")
for extra in sys.argv[2:]:
    k, v = extra.split("="); env[k] = v
t = time.time()
subprocess.run([os.path.join(ROOT, "pintron_amd", "bin", "est-fact")], cwd=d, env=env, check=True)
dt = time.time() - t
aligned = sum(1 for l in open(os.path.join(d, "processed-ests.txt")) if l.startswith(">"))
print("est-fact: %d ESTs in %.2fs => %.0f ESTs/s (aligned %d)" % (n, dt, n / dt, aligned), flush=True)
