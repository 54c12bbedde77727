#!/bin/bash
# A/B of host-library variants under build_ab/ on one box:  bash tools/ab.sh [variant ...]
run() { echo "$*"; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step']), {k:round(v,2) for k,v in d.get('phases_s',{}).items()})
print('  ', [(k['name'], round(k.get('ms_per_step', 0),1)) for k in d.get('kernels',[]) if k['name'].startswith('pair')])"; }
for rep in 1 2; do
  run X=default
  for v in "$@"; do run PINTRON_ESTFACT_LIB=$PWD/build_ab/libestfact_$v.so; done
done
