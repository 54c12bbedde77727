#!/bin/bash
# GPU box: bench.py bound to the CPUs of the GPU's NUMA node, to the other node, and unbound.
N=$(rocm-smi --showtopo 2>/dev/null | awk '/Numa Node:/ {print $NF; exit}')
near=$(cat /sys/devices/system/node/node$N/cpulist)
other=$(cat /sys/devices/system/node/node$((1-N))/cpulist)
echo "GPU on NUMA node $N: near cpus $near, other $other"
run() { "$@" python bench.py --steps 5 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step']), round(d['phases_s']['host_cpu_per_thread'],3), round(d['phases_s']['dp_batches_per_thread'],3))"; }
for r in 1 2 3; do
  echo unbound; run env X=1
  echo near; run taskset -c $near
  echo other; run taskset -c $other
done
