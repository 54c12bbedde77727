export TMPDIR=/tmp
for cfg in "X=1" "PINTRON_SERVICES=1" "X=2"; do
  rm -rf gpurun_out/rp_test
  echo "== $cfg"
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_test -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/rp_test.json 2> gpurun_out/rp_test.err; echo "rc=$?"; grep -A22 "SIGSEGV" gpurun_out/rp_test.err | head -30
done
find gpurun_out/rp_test -name '*kernel_trace.csv' -delete
