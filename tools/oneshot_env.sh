#!/bin/bash
# GPU box: one est-fact process on the C3 batch under the given environment settings, twice each
W=/tmp/oneshot_c3; mkdir -p $W
[ -f $W/ests.txt ] || python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("C3", seed=3), "$W")
PY
cd $W
for cfg in "$@"; do
  for i in 1 2; do
    T0=$(date +%s.%N); env $cfg PINTRON_VERBOSE=1 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2>&1 | grep "^\* run\|^\* cpu" | cut -c1-200; T1=$(date +%s.%N)
    python3 -c "print('$cfg wall %.3f s' % ($T1 - $T0))"
  done
done
md5sum raw-multifasta-out.txt
