#!/bin/bash
W=/tmp/ab; mkdir -p $W
python3 -c "
import sys; sys.path.insert(0, '.')
from pintron_amd import synth
synth.write_files(synth.make('C3', seed=3, n_est=20000), '$W')"
cd $W
for E in A=1 PINTRON_GPU_MEG=0 PINTRON_SERVICES=1 PINTRON_SERVICES=2 PINTRON_NO_PREFETCH=1 PGPU_MERGED=0 GPU_MAX_HW_QUEUES=1 "GPU_MAX_HW_QUEUES=1 PINTRON_SERVICES=1"; do
  env $E PINTRON_VERBOSE=2 $GRAFT_REPO_ROOT/pintron_amd/bin/est-fact 2> err.txt
  echo "$E: $(grep -c '17[0-9] MB of' err.txt) areas; $(grep 'resident at' err.txt)"
done
