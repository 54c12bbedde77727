#!/bin/bash
# GPU box: bench.py with its threads bound to (a) the GPU's NUMA node (the product's default),
# (b) one hardware thread per core of that node, (c) unbound.
N=$(rocm-smi --showtopo 2>/dev/null | awk '/Numa Node:/ {print $NF; exit}')
near=$(cat /sys/devices/system/node/node$N/cpulist)
first=${near%%,*}
echo "GPU on NUMA node $N: cpus $near, first hardware threads $first"
run() { "$@" python bench.py --steps 5 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step']), round(d['phases_s']['host_cpu_per_thread'],3), round(d['phases_s']['dp_batches_per_thread'],3))"; }
for r in 1 2 3; do
  echo node; run env X=1
  echo cores; run taskset -c $first
  echo unbound; run env PINTRON_NUMA=0
done
