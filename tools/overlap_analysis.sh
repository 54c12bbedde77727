#!/bin/bash
# GPU box: do the pairing / MEG kernels of the prefetch thread slow the DP batches that run beside them?
# kernel trace of a short bench, then dp_batch / lcf durations split by "a prefetch kernel was running".
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/trace_ov; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_ov -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-oneshot > gpurun_out/trace_ov.json 2> gpurun_out/trace_ov.err || { tail -3 gpurun_out/trace_ov.err; exit 1; }
python3 - <<'PY'
import csv, glob, bisect, statistics
f = glob.glob("gpurun_out/trace_ov/**/*kernel_trace.csv", recursive=True)[0]
K = list(csv.DictReader(open(f)))
pre = sorted((int(k["Start_Timestamp"]), int(k["End_Timestamp"])) for k in K if any(x in k["Kernel_Name"] for x in ("pair_", "meg_", "scan_")))
starts = [p[0] for p in pre]
def overlapped(s, e):
    i = bisect.bisect_right(starts, e) - 1
    while i >= 0 and pre[i][0] > s - 5_000_000:
        if pre[i][1] > s and pre[i][0] < e: return True
        i -= 1
    return False
for name in ("dp_batch_kernel", "lcf_kernel"):
    a, b = [], []
    for k in K:
        if name not in k["Kernel_Name"]: continue
        s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
        (a if overlapped(s, e) else b).append((e - s) / 1000.0)
    for tag, v in (("beside prefetch kernels", a), ("alone", b)):
        if v: print("%-16s %-24s n=%5d  mean %7.1f us  median %7.1f us" % (name, tag, len(v), statistics.mean(v), statistics.median(v)))
PY
find gpurun_out/trace_ov -name '*.csv' -size +1M -delete
