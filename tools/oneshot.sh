#!/bin/bash
# GPU box: wall-clock of the est-fact PROCESS (start to files on disk) on a C3 batch
N=${1:-100000}
W=$(mktemp -d)
python - "$W" "$N" <<'PY'
import sys
sys.path.insert(0, '.')
from pintron_amd import synth
synth.write_files(synth.make("C3", n_est=int(sys.argv[2])), sys.argv[1])
PY
cd $W
for i in 1 2; do
  s=$(date +%s.%N)
  PINTRON_VERBOSE=1 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2>&1 | grep "est-fact:\|run:" | tail -2
  e=$(date +%s.%N)
  python3 -c "print('process wall %.2f s -> %.0f ESTs/s' % ($e-$s, $N/($e-$s)))"
done
ls -la raw-multifasta-out.txt megs.txt | awk '{print $5, $9}'
