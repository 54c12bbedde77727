#!/usr/bin/env python3
"""Which kernel finishes last in each merged DP batch?  Reads a rocprofv3 kernel trace
(tools/trace_batch.sh -> gpurun_out/trace1/*kernel_trace.csv), groups the dispatches by launching
thread into batches (a gap of more than 150 us between consecutive dispatches of one thread starts
a new batch) and prints, per kernel, how often it was the long pole and the mean batch span."""
import csv, glob, re, sys, collections
path = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/trace1/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
by_thread = collections.defaultdict(list)
for r in rows:
    by_thread[r["Thread_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "")
pole = collections.Counter(); pole_ms = collections.defaultdict(float); spans = []; tails = collections.defaultdict(list)
for t, ks in by_thread.items():
    ks.sort()
    if not any("any_kernel" in k[2] or "lev_wave" in k[2] for k in ks):
        continue
    batch = []
    def close(b):
        if len(b) < 3: return
        s0 = min(k[0] for k in b); e1 = max(k[1] for k in b)
        last = max(b, key=lambda k: k[1])
        second = sorted(k[1] for k in b)[-2]
        spans.append((e1 - s0) / 1e3)
        pole[short(last[2])] += 1; pole_ms[short(last[2])] += (last[1] - last[0]) / 1e3
        tails[short(last[2])].append((e1 - second) / 1e3)
    prev = None
    for k in ks:
        if prev is not None and k[0] - prev > 150000:
            close(batch); batch = []
        batch.append(k); prev = k[0]
    close(batch)
spans.sort()
print("batches %d; span us: mean %.0f median %.0f p90 %.0f" % (len(spans), sum(spans) / len(spans), spans[len(spans) // 2], spans[int(len(spans) * .9)]))
for n, c in pole.most_common():
    tl = tails[n]
    print("%-40s last in %4d batches; its mean duration %.0f us; finishes %.0f us after the runner-up" % (n, c, pole_ms[n] / c, sum(tl) / len(tl)))
