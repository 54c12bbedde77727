#!/bin/bash
# GPU box: one-shot est-fact under two environments, alternating, in a FRESH directory every run (as the
# pipeline driver runs it); usage: oneshot_ab.sh WORKLOAD "ENV_A" "ENV_B" [runs]
W=/tmp/oneshot_ab_in; mkdir -p $W
python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("$1", seed=3), "$W")
PY
for i in $(seq 1 ${4:-3}); do
  for E in "$2" "$3"; do
    D=/tmp/oneshot_ab_run; rm -rf $D; mkdir -p $D; ln $W/genomic.txt $W/ests.txt $D/ 2>/dev/null || cp $W/genomic.txt $W/ests.txt $D/
    cd $D
    T0=$(date +%s.%N); env $E PINTRON_VERBOSE=1 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2> err.txt; T1=$(date +%s.%N)
    python3 - "$T0" "$T1" "$E" <<'PY'
import re, sys
t0, t1 = float(sys.argv[1]), float(sys.argv[2])
s = open("err.txt").read()
b = float(re.search(r"main leaves at ([0-9.]+)", s).group(1))
print("%-28s wall %.3f (after main %.3f) | %s" % (sys.argv[3][:28], t1 - t0, t1 - b, re.search(r"\* run: (.*)", s).group(1)))
PY
    cd /tmp
  done
done
md5sum /tmp/oneshot_ab_run/raw-multifasta-out.txt
