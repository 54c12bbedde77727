#!/bin/bash
W=/tmp/ab; mkdir -p $W
python3 -c "
import sys; sys.path.insert(0, '.')
from pintron_amd import synth
synth.write_files(synth.make('C3', seed=3, n_est=20000), '$W')"
cd $W
AMD_LOG_LEVEL=4 PINTRON_VERBOSE=2 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2> err.txt
grep '17[0-9] MB of' err.txt
for a in $(grep '17[0-9] MB of' err.txt | sed -e 's/.*anon \([0-9a-f]*\) .*/\1/'); do
  echo "== $a"; grep -i "$a" err.txt | grep -v "MB of" | sed -e "s/[0-9]* us: \[pid:[0-9]* tid: 0x[0-9a-f]*\]//" | cut -c1-260 | sort | uniq -c | sort -rn | head -4
done
