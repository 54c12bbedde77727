"""GPU box: does the CPU quota of the box (cgroup v2 cpu.max) throttle a step?  Prints cpu.stat's deltas over
warm C3 steps under the given PINTRON_* settings (usage, nr_periods, nr_throttled, throttled_usec)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pintron_amd import synth
from pintron_amd.estfact import Session, load_host_lib


def cpu_stat():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.stat",):
        try:
            for ln in open(p):
                k, v = ln.split()
                out[k] = int(v)
        except OSError:
            pass
    return out


L = load_host_lib()
d = tempfile.mkdtemp()
synth.write_files(synth.make(os.environ.get("WORKLOAD", "C3")), d)
s = Session(L, d)
for _ in range(3):
    s.step()
n = int(os.environ.get("STEPS", "10"))
a = cpu_stat(); t0 = time.perf_counter()
for _ in range(n):
    s.step()
t1 = time.perf_counter(); b = cpu_stat()
s.close()
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
print("%d steps, %.1f ms per step" % (n, (t1 - t0) / n * 1e3))
for k in ("usage_usec", "user_usec", "system_usec", "nr_periods", "nr_throttled", "throttled_usec"):
    if k in a:
        print("  %-16s %12d  (%.1f per step)" % (k, b[k] - a[k], (b[k] - a[k]) / n))
