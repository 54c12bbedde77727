#!/bin/bash
# GPU box: default bench line, then the same command under rocprofv3 (kernel trace + stats).
# Results under gpurun_out/; the summaries worth keeping are copied into profiles/ by hand.
#   bash tools/profile_bench.sh [WORKLOAD]     (C3 by default; C5, C2, "C4 --genes 1")
set -e
W=${1:-C3}
TAG=$(echo $W | tr -d " -" | tr 'A-Z' 'a-z')
mkdir -p gpurun_out
python bench.py --workload $W > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err || { tail -5 gpurun_out/bench_default_$TAG.err; exit 1; }
tail -1 gpurun_out/bench_default_$TAG.json | cut -c1-600
export TMPDIR=/tmp
# rocprofv3 itself crashes now and then on this workload (SIGSEGV inside its interception of a HIP
# call, seen in about one profiled run out of ten, never without the profiler): try up to 3 times
for attempt in 1 2 3; do
  rm -rf gpurun_out/prof_bench
  if rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --workload $W --steps 5 --warmup 1 --no-cpu --no-oneshot --no-extra > gpurun_out/bench_under_rocprof_$TAG.json 2> gpurun_out/rocprof.err; then break; fi
  echo "rocprofv3 attempt $attempt failed"; tail -3 gpurun_out/rocprof.err
  [ $attempt = 3 ] && exit 1
done
find gpurun_out/prof_bench -name '*kernel_stats.csv' | while read f; do cp "$f" gpurun_out/bench_kernel_stats_$TAG.csv; done
find gpurun_out/prof_bench -name '*kernel_trace.csv' -delete
head -12 gpurun_out/bench_kernel_stats_$TAG.csv
