#!/bin/bash
# GPU box: host-side load of a full C5 input (1 Mb, two million reads) by phases
W=/tmp/lt; mkdir -p $W
python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("${1:-C5}", seed=3), "$W")
PY
cd $W
ls -la ests.txt
for t in 8 16; do
  $GRAFT_REPO_ROOT/tools/exp/load_time $t
  PINTRON_ARENA_THP=1 $GRAFT_REPO_ROOT/tools/exp/load_time $t 2>&1 | sed -e 's/^/THP /'
done
