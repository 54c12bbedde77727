#!/usr/bin/env python3
"""CPU seconds per thread of one process running C3 steps (GPU box): who uses the host share?
Worker, service and prefetch threads end with each step, so their time shows up in the process
total; the long-lived threads (HIP runtime helpers, the Python main thread) are listed by name."""
import os, sys, time, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (same load order as bench.py)
from pintron_amd import synth
from pintron_amd.estfact import Session, load_host_lib

def snapshot():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"), int(rest[12]) / os.sysconf("SC_CLK_TCK"))
        except OSError:
            pass
    return out

import threading, collections
seen = {}
stop = False
def sampler():
    while not stop:
        for tid, v in snapshot().items():
            seen[tid] = v
        time.sleep(0.01)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
work = tempfile.mkdtemp(prefix="pintron_tcpu_")
synth.write_files(synth.make("C3"), work)
s = Session(load_host_lib(), work)
s.step()
seen.clear(); th = threading.Thread(target=sampler); th.start()
a = snapshot(); t0 = time.perf_counter(); c0 = time.process_time()
for _ in range(steps):
    s.step()
wall = time.perf_counter() - t0; cpu = time.process_time() - c0
b = snapshot()
stop = True; th.join()
by_name = collections.defaultdict(lambda: [0, 0.0, 0.0])
for tid, (name, sec, sys_s) in seen.items():
    base = a[tid][1] if tid in a else 0.0
    base_sys = a[tid][2] if tid in a else 0.0
    by_name[name][0] += 1; by_name[name][1] += sec - base; by_name[name][2] += sys_s - base_sys
print("sampled (10 ms; short-lived threads lose their last slice):")
for name, (n, sec, sys_s) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
    if sec / steps > 0.002:
        print("  %-16s %5d threads  %.3f s/step (of which %.3f in the kernel)" % (name, n, sec / steps, sys_s / steps))
print("steps %d: wall %.3f s/step, process cpu %.3f s/step (%.1f cores busy)" % (steps, wall / steps, cpu / steps, cpu / wall))
for tid, (name, sec, _sys) in sorted(b.items(), key=lambda kv: -kv[1][1]):
    d = sec - a.get(tid, (name, 0.0, 0.0))[1]
    if d > 0.005 * steps:
        print("  long-lived thread %-18s %s %.3f s/step" % (name, "(main)" if tid == os.getpid() else "(tid %d, not main)" % tid, d / steps))
s.close(); shutil.rmtree(work, ignore_errors=True)
