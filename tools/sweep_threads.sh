#!/bin/bash
# GPU box: bench.py (C3) with 8 .. 48 worker threads, on one hardware thread per core of the GPU's NUMA
# node (the default binding) and on every hardware thread of it (PINTRON_SMT=1); three alternating rounds.
echo "nproc $(nproc), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), affinity $(taskset -pc $$ | sed 's/.*: //')"
lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|NUMA node" | sed 's/^/  /'
thr() { awk '/nr_throttled|throttled_usec/ {printf "%s ", $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null; }
run() { local t0=($(thr)); env "$@" python bench.py --steps 6 --warmup 3 --no-cpu --no-oneshot 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['phases_s']; print('   %7d ESTs/s  %6.1f ms  host/thread %.3f  dp-wait/thread %.3f' % (d['value'], d['ms_per_step'], p['host_cpu_per_thread'], p['dp_batches_per_thread']))"; local t1=($(thr)); [ -n "${t1[0]}" ] && echo "      cgroup throttled: $((t1[0]-t0[0])) periods, $(( (t1[1]-t0[1]) / 1000 )) ms (whole bench process incl. load)"; }
for r in 1 2 3; do
  for t in ${SWEEP_THREADS:-8 16 24 32 48}; do
    echo "round $r threads $t cores"; run PINTRON_THREADS=$t
    echo "round $r threads $t smt";   run PINTRON_THREADS=$t PINTRON_SMT=1
  done
done
