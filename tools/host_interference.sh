#!/bin/bash
# GPU box with ONE GPU: bench.py with 1, 2 and 4 ranks that SHARE the GPU (exchange over gloo: two RCCL
# ranks cannot sit on one device).  What it measures is the host side of weak scaling -- N sessions' worker
# threads, service threads and page-locked buffers on one host, under one CPU quota -- not xGMI.  Every rank
# runs a C3 batch of $1 ESTs (default 25 000) so that the GPU itself is far from full at N = 8.
E=${1:-25000}
echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) nproc $(nproc)"
for n in 1 2 4; do      # the pool allows six processes on a GPU: eight ranks cannot share one here
  PINTRON_DIST_BACKEND=gloo python bench.py --gpus $n --ests $E --steps 5 --warmup 2 --no-cpu --no-oneshot 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['phases_s']
print('ranks %d: %8d input ESTs/s whole job, %6.1f ms/step, host/thread %.3f s, dp-wait/thread %.3f s, threads/rank %s' % (d['n_gpus'], d['input_ests_per_s'], d['ms_per_step'], p['host_cpu_per_thread'], p['dp_batches_per_thread'], d['config']['stages'].split('(')[1].split(' ')[0]))"
done
