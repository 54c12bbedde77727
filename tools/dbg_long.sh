#!/bin/bash
# debugging aid: the same input through est-fact many times; every run must equal the reference
W=$(mktemp -d)
python - "$W" <<'PY'
import sys, os
sys.path.insert(0, '.')
from pintron_amd import synth
g, e = synth.make_long_transcripts()
for d in ("ref", "mine"):
    os.makedirs(os.path.join(sys.argv[1], d))
    open(os.path.join(sys.argv[1], d, "genomic.txt"), "w").write(g)
    open(os.path.join(sys.argv[1], d, "ests.txt"), "w").write(e)
PY
(cd $W/ref && $OLDPWD/oracle/_ref/est-fact-core 2>/dev/null)
bad=0
for k in $(seq 1 ${1:-25}); do
  (cd $W/mine && $OLDPWD/pintron_amd/bin/est-fact 2>/dev/null)
  for f in raw-multifasta-out.txt megs.txt processed-megs.txt meg-edges.txt processed-ests.txt; do
    if ! cmp -s $W/mine/$f $W/ref/$f; then echo "run $k: $f DIFF"; bad=$((bad+1)); cp $W/mine/$f gpurun_out/dbg_run${k}_$f; fi
  done
done
cp $W/ref/raw-multifasta-out.txt gpurun_out/dbg_ref_raw.txt; cp $W/ref/megs.txt gpurun_out/dbg_ref_megs.txt
echo "mismatching files over all runs: $bad"
