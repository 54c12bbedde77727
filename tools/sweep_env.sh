#!/bin/bash
# bench.py under several environment settings on one box:  bash tools/sweep_env.sh "A=1 B=2" "A=2" ...
for cfg in "$@"; do
  echo "$cfg"
  env $cfg python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step']), {k:round(v,2) for k,v in d.get('phases_s',{}).items()}, 'launches/step', round(sum(k.get('launches',0) for k in d.get('kernels',[]))))"
done
