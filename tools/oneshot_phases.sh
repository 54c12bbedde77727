#!/bin/bash
# GPU box: phases of one est-fact process on the C3 batch (the `oneshot` of bench.py), verbose.
W=/tmp/oneshot_c3; mkdir -p $W
python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("C3", seed=3), "$W")
PY
cd $W
for i in 1 2 3; do
  T0=$(date +%s.%N); PINTRON_VERBOSE=1 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2>&1 | grep -v "^\* service" | tail -6; T1=$(date +%s.%N); python3 -c "print(\"wall %.3f s\" % ($T1 - $T0))"
done
