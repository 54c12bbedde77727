run() { env "$@" python bench.py --workload $W --steps 8 --warmup 3 --no-cpu --no-oneshot 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['phases_s']; print('   %7d in/s  %6.1f ms  host %.3f  dp %.3f  batches %d' % (d['input_ests_per_s'], d['ms_per_step'], p['host_cpu_per_thread'], p['dp_batches_per_thread'], d['config']['dp_batches_per_step']))"; }
for W in C5 "C4 --genes 1"; do for r in 1 2; do
for cfg in "X=1" "PINTRON_SERVICES=4 PINTRON_COALESCE_US=0 PINTRON_LANES=8" "PINTRON_SERVICES=4 PINTRON_COALESCE_US=0 PINTRON_LANES=8 PINTRON_FIBERS=768"; do
  echo "$W round $r: $cfg"; run $cfg
done; done; done
