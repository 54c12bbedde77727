#!/bin/bash
# GPU box: flat gprof profile of the product est-fact (host side) on a C3 batch.
set -e
N=${1:-100000}
W=$(mktemp -d)
python - "$W" "$N" <<'PY'
import sys
sys.path.insert(0, '.')
from pintron_amd import synth
w = synth.make("C3", n_est=int(sys.argv[2]))
synth.write_files(w, sys.argv[1])
PY
gcc -std=gnu99 -O2 -pg -fno-ipa-sra -fno-ipa-cp -fno-partial-inlining -pthread -o $W/est-fact-pg pintron_amd/host/*.c -Lpintron_amd/lib -lpintron_gpu -lm -Wl,-rpath,$PWD/pintron_amd/lib -Wl,-rpath-link,/opt/rocm/lib
( cd $W && PINTRON_VERBOSE=1 ./est-fact-pg 2>&1 | tail -3 && gprof -b -p ./est-fact-pg gmon.out | head -60 ) > gpurun_out/gprof_product.txt 2>&1
tail -70 gpurun_out/gprof_product.txt
