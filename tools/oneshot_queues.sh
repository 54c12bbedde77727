#!/bin/bash
# GPU box: what the number of hardware queues (GPU_MAX_HW_QUEUES) costs a one-shot est-fact process and gives a
# steady-state step.  Every HSA queue has a context-save area for all 256 CUs in host memory (173 MB on this
# device) that is page-faulted in when the queue is made and taken apart when the process ends.
W=/tmp/oneshot_q; mkdir -p $W
python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("${1:-C3}", seed=3), "$W")
PY
cd $W
for Q in default 1 2 4 8; do
  for i in 1 2; do
    if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
    T0=$(date +%s.%N); PINTRON_VERBOSE=2 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2> err.txt; T1=$(date +%s.%N)
    python3 - "$T0" "$T1" "$Q" <<'PY'
import re, sys
t0, t1 = float(sys.argv[1]), float(sys.argv[2])
s = open("err.txt").read()
a = float(re.search(r"main entered at ([0-9.]+)", s).group(1)); b = float(re.search(r"main leaves at ([0-9.]+)", s).group(1))
run = re.search(r"\* run: (.*)", s).group(1)
res = re.search(r"resident at the end of main: (\d+) MB", s).group(1)
big = len(re.findall(r"^\*\s+17\d MB of\s+17\d\s+\[anon", s, re.M))
print("queues %-7s wall %.3f s = before main %.3f + main %.3f + after main %.3f | %s | resident %s MB, %d+ areas of 173 MB" % (sys.argv[3], t1 - t0, a - t0, b - a, t1 - b, run, res, big))
PY
  done
done
cd $GRAFT_REPO_ROOT
for Q in default 2 4 8; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  echo "bench queues $Q: $(python3 bench.py --steps 10 --warmup 3 --no-oneshot --no-cpu 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["input_ests_per_s"], d["ms_per_step"])')"
done
