#!/bin/bash
# GPU box: fibres per worker x lanes per worker, steady state, C3 (and C5 share)
for W in ${1:-C3}; do
for F in 384 512 768 1024 1536; do
  for L in 4 8 16; do
    PINTRON_FIBERS=$F PINTRON_LANES=$L python3 bench.py --workload $W --no-cpu --no-oneshot --steps 12 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_s']
print('$W fibres %5d lanes %2d: %8d input ESTs/s %7.2f ms  host/thread %.4f dp/thread %.4f' % ($F, $L, d['input_ests_per_s'], d['ms_per_step'], p['host_cpu_per_thread'], p['dp_batches_per_thread']))"
  done
done
done
