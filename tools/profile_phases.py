"""GPU box: PINTRON_PROFILE=1 phase table of warm C3 steps (per worker thread), the scheduler's rows first."""
import os, sys, tempfile, re, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from pintron_amd import synth
    from pintron_amd.estfact import Session, load_host_lib
    d = tempfile.mkdtemp()
    synth.write_files(synth.make(os.environ.get("WORKLOAD", "C3")), d)
    s = Session(load_host_lib(), d)
    for k in range(5):
        sys.stderr.write("=== step %d\n" % k); sys.stderr.flush()
        s.step()
    s.close()
    sys.exit(0)
e = dict(os.environ, PINTRON_PROFILE="1")
r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, stderr=subprocess.PIPE, text=True)
steps = r.stderr.split("=== step ")[3:]          # the third step on: warm
acc = {}
for st in steps:
    for m in re.finditer(r"^\*\s+(\S[^\n]*?)\s+([\d.]+) s\s+[\d.]+\s+[\d.]+$", st, re.M):
        acc.setdefault(m.group(1), []).append(float(m.group(2)))
keys = ["sched:launch", "sched:collect", "sched:start", "scheduler", "sleep", "wait:prefetch", "output", "side-files", "refine-intron", "all but sleeps"]
print("  ".join("%s %.4f" % (k, sum(acc[k]) / len(acc[k])) for k in keys if k in acc))
