#!/bin/bash
# GPU box: bench.py (C3) under scheduler settings, two alternating rounds (PINTRON_* / PGPU_* environment)
run() { env "$@" python bench.py --workload ${SWEEP_WORKLOAD:-C3} --steps 8 --warmup 3 --no-cpu --no-oneshot --no-extra 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['phases_s']; print('   %7d in/s  %6.1f ms  host %.3f  dp %.3f  batches %d' % (d['input_ests_per_s'], d['ms_per_step'], p['host_cpu_per_thread'], p['dp_batches_per_thread'], d['config']['dp_batches_per_step']))"; }
for r in 1 2; do
for cfg in ${SWEEP_CFGS:-"X=1"}; do
  c=$(echo $cfg | tr ',' ' ')
  echo "round $r: $c"; run $c
done; done
