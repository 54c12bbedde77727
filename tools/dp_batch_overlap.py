#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: how long does dp_batch_kernel run as a function of its size (workgroups) and
of what else is on the chip while it runs (other dp_batch launches, pairing / MEG kernels)?
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-oneshot
  python tools/dp_batch_overlap.py gpurun_out/tr"""
import csv, glob, sys, collections
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tr"
K = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
ev = [(int(k["Start_Timestamp"]), int(k["End_Timestamp"]), k["Kernel_Name"], int(k.get("Grid_Size_X", k.get("Grid_Size", 0)) or 0), int(k.get("Workgroup_Size_X", 512) or 512)) for k in K]
dp = [e for e in ev if "dp_batch_kernel" in e[2]]
oth = [e for e in ev if "dp_batch_kernel" not in e[2]]
print("dp_batch launches %d, mean %.1f us" % (len(dp), sum(e[1] - e[0] for e in dp) / len(dp) / 1e3))
dp.sort()
import bisect
starts = [e[0] for e in dp]
def overlap_count(e, pool):
    return sum(1 for o in pool if o is not e and o[0] < e[1] and o[1] > e[0])
by_wg = collections.defaultdict(list); by_ov = collections.defaultdict(list); by_oth = collections.defaultdict(list)
for i, e in enumerate(dp):
    lo = bisect.bisect_left(starts, e[0] - 3_000_000); hi = bisect.bisect_right(starts, e[1])
    n_ov = sum(1 for o in dp[lo:hi] if o is not e and o[0] < e[1] and o[1] > e[0])
    wgs = e[3] // max(e[4], 1)
    dur = (e[1] - e[0]) / 1e3
    by_wg[min(wgs // 100, 12)].append(dur); by_ov[min(n_ov, 6)].append(dur)
    big = any(o[0] < e[1] and o[1] > e[0] for o in oth if ("pair_" in o[2] or "meg_" in o[2] or "scan_" in o[2]))
    by_oth[big].append(dur)
print("by workgroups (x100):")
for k in sorted(by_wg): v = by_wg[k]; print("  %4d..%4d wgs: n %5d mean %6.1f us  median %6.1f" % (k * 100, k * 100 + 99, len(v), sum(v) / len(v), sorted(v)[len(v) // 2]))
print("by number of other dp_batch kernels overlapping in time:")
for k in sorted(by_ov): v = by_ov[k]; print("  %d%s: n %5d mean %6.1f us  median %6.1f" % (k, "+" if k == 6 else "", len(v), sum(v) / len(v), sorted(v)[len(v) // 2]))
print("while a pairing / MEG / scan kernel runs:")
for k in (False, True):
    v = by_oth[k]
    if v: print("  %s: n %5d mean %6.1f us  median %6.1f" % (k, len(v), sum(v) / len(v), sorted(v)[len(v) // 2]))
