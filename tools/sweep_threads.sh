#!/bin/bash
# usage: bash tools/sweep_threads.sh  -- bench.py under different scheduler settings (GPU box)
run() { echo "$*"; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step']), {k:round(v,2) for k,v in d.get('phases_s',{}).items()})
if '$SHOWK': print('  ', [(k['name'], round(k['ms'],1), k['jobs']) for k in d.get('kernels',[])])"; }
SHOWK=
SWEEP=${SWEEP:-32:256 48:256 32:512 24:512 48:128 64:128 24:1024}
for TF in $SWEEP; do run PINTRON_THREADS=${TF%%:*} PINTRON_FIBERS=${TF##*:}; done
