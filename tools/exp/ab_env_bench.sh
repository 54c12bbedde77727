#!/bin/bash
# GPU box: steady-state bench under two environments, alternating; usage: ab_env_bench.sh WORKLOAD "ENV_A" "ENV_B" [rounds]
for i in $(seq 1 ${4:-3}); do
  for E in "$2" "$3"; do
    env $E python3 bench.py --workload $1 --no-cpu --no-oneshot --steps 15 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_s']
print('%-24s %s %8d input ESTs/s %7.2f ms  host/thread %.4f dp/thread %.4f' % ('$E', '$1', d['input_ests_per_s'], d['ms_per_step'], p['host_cpu_per_thread'], p['dp_batches_per_thread']))"
  done
done
