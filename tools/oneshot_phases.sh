#!/bin/bash
# GPU box: phases of one est-fact process on the C3 batch (the `oneshot` of bench.py), verbose: before main
# (loader), open / step / write, and after main (the kernel's teardown of the process).
W=/tmp/oneshot_c3; mkdir -p $W
python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("${1:-C3}", seed=3), "$W")
PY
cd $W
for i in 1 2 3; do
  T0=$(date +%s.%N); PINTRON_VERBOSE=${VERBOSE:-1} $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2> err.txt; T1=$(date +%s.%N)
  grep -v "^\* service" err.txt | tail -${TAIL:-8}
  python3 - "$T0" "$T1" <<'PY'
import re, sys
t0, t1 = float(sys.argv[1]), float(sys.argv[2])
s = open("err.txt").read()
a = float(re.search(r"main entered at ([0-9.]+)", s).group(1)); b = float(re.search(r"main leaves at ([0-9.]+)", s).group(1))
print("wall %.3f s = before main %.3f + main %.3f + after main %.3f" % (t1 - t0, a - t0, b - a, t1 - b))
PY
done
