#!/bin/bash
# GPU box: one-shot est-fact on C3 under a few environments (each twice)
W=/tmp/oneshot_e; mkdir -p $W
python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("${1:-C3}", seed=3), "$W")
PY
cd $W
one() {
  T0=$(date +%s.%N); env "$@" PINTRON_VERBOSE=2 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2> err.txt; T1=$(date +%s.%N)
  python3 - "$T0" "$T1" "$*" <<'PY'
import re, sys
t0, t1 = float(sys.argv[1]), float(sys.argv[2])
s = open("err.txt").read()
a = float(re.search(r"main entered at ([0-9.]+)", s).group(1)); b = float(re.search(r"main leaves at ([0-9.]+)", s).group(1))
run = re.search(r"\* run: (.*)", s).group(1)
res = re.search(r"resident at the end of main: (.*)", s).group(1)
pf = re.findall(r"(\d+) page faults", s)
print("%-44s wall %.3f = main %.3f + after %.3f | %s | %s | faults %s" % (sys.argv[3][:44], t1 - t0, b - a, t1 - b, run, res, "+".join(pf)))
PY
}
for i in 1 2 3 4; do
  one PINTRON_WARM_ARENAS=1
  one PINTRON_WARM_ARENAS=0
done
md5sum raw-multifasta-out.txt
