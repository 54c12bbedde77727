#!/bin/bash
# GPU box: one est-fact process on a synthetic config:  bash tools/oneshot_cfg.sh C5 200000
CFG=${1:-C5}; N=${2:-200000}
W=$(mktemp -d)
python - "$W" "$CFG" "$N" <<'PY'
import sys, time
sys.path.insert(0, '.')
from pintron_amd import synth
t=time.time()
synth.write_files(synth.make(sys.argv[2], n_est=int(sys.argv[3])), sys.argv[1])
print("generated in %.1f s" % (time.time()-t))
PY
cd $W
s=$(date +%s.%N)
PINTRON_VERBOSE=1 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2>&1 | grep "est-fact:\|run:\|FATAL" | cut -c1-260
e=$(date +%s.%N)
python3 -c "print('process wall %.2f s -> %.0f ESTs/s' % ($e-$s, $N/($e-$s)))"
grep -c ">" raw-multifasta-out.txt
