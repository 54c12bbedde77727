#!/usr/bin/env python3
"""GPU-side timeline of the merged DP batches from a rocprofv3 trace (tools/trace_batch.sh):
upload -> kernels -> download per batch, with the mean of every gap, and the hardware queue each
kernel family landed on.  Streams of one context are consecutive ids (main, then the auxiliaries)."""
import csv, glob, sys, collections, re, bisect
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace1"
K = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
M = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "")
dp = [k for k in K if any(x in k["Kernel_Name"] for x in ("any_kernel", "lev_wave", "lcf_kernel"))]
qs = collections.defaultdict(collections.Counter)
for k in dp: qs[short(k["Kernel_Name"])][k["Queue_Id"]] += 1
print("hardware queue of each kernel (queue id: launches):")
for n, c in qs.items(): print("  %-34s %s" % (n, dict(c)))
# contexts: kernel streams grouped by launching thread
streams_of = collections.defaultdict(set)
for k in dp: streams_of[k["Thread_Id"]].add(int(k["Stream_Id"]))
for t, ss in streams_of.items():
    copy_streams = {int(m["Stream_Id"]) for m in M}
    main = min(ss) if min(ss) in copy_streams else min(ss) - 1
    ks = sorted((int(k["Start_Timestamp"]), int(k["End_Timestamp"])) for k in dp if k["Thread_Id"] == t)
    h2d = sorted((int(m["Start_Timestamp"]), int(m["End_Timestamp"])) for m in M if int(m["Stream_Id"]) == main and "HOST_TO_DEVICE" in m["Direction"])
    d2h = sorted((int(m["Start_Timestamp"]), int(m["End_Timestamp"])) for m in M if int(m["Stream_Id"]) == main and "DEVICE_TO_HOST" in m["Direction"])
    if not h2d or not d2h: continue
    starts = [k[0] for k in ks]
    rows = []
    for i, (us, ue) in enumerate(h2d):
        nxt = h2d[i + 1][0] if i + 1 < len(h2d) else 1 << 62
        lo = bisect.bisect_left(starts, us); hi = bisect.bisect_left(starts, nxt)
        if hi - lo < 3: continue
        kb = ks[lo:hi]
        dd = [x for x in d2h if us < x[0] < nxt]
        if not dd: continue
        k0 = min(x[0] for x in kb); k1 = max(x[1] for x in kb)
        rows.append((ue - us, k0 - ue, k1 - k0, dd[-1][0] - k1, dd[-1][1] - dd[-1][0], dd[-1][1] - us, nxt - dd[-1][1] if nxt < 1 << 62 else 0))
    if not rows: continue
    n = len(rows)
    mean = [sum(r[c] for r in rows) / n / 1e3 for c in range(7)]
    print("thread %s (streams %d..%d): %d batches; us: upload %.0f, gap %.0f, kernels %.0f, gap %.0f, download %.0f, total %.0f; until the next upload %.0f"
          % (t, main, max(ss), n, *mean))
