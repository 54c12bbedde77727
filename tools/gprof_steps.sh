#!/bin/bash
# GPU box: flat gprof profile of the host side over WARM steps of the real product code (session API,
# real GPU library): tests/hostcheck/sched_profile_main.c linked with pintron_amd/host/*.c, -pg.
# Samples land on whichever thread is running when the process-wide profiling timer fires.
set -e
W=/tmp/gprof_c3; mkdir -p $W gpurun_out
[ -f $W/ests.txt ] || python3 - <<PY
import sys; sys.path.insert(0, ".")
from pintron_amd import synth
synth.write_files(synth.make("C3", seed=3), "$W")
PY
SRC=$(ls pintron_amd/host/*.c | grep -v est_fact_main.c)
gcc -std=gnu99 -O2 -pg -fno-ipa-sra -fno-ipa-cp -fno-partial-inlining -pthread -o $W/steps-pg tests/hostcheck/sched_profile_main.c $SRC \
    -Lpintron_amd/lib -lpintron_gpu -lm -Wl,-rpath,$PWD/pintron_amd/lib -Wl,-rpath-link,/opt/rocm/lib
( cd $W && ./steps-pg ${1:-6} 2>&1 | tail -7 && gprof -b -p ./steps-pg gmon.out | head -70 ) > gpurun_out/gprof_steps.txt 2>&1
head -80 gpurun_out/gprof_steps.txt
