export TMPDIR=/tmp
fails=0
for it in 1 2 3 4 5 6 7 8; do
  rm -rf gpurun_out/rp_test
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_test -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/rp_test.json 2> gpurun_out/rp_test.err; rc=$?
  echo "run $it rc=$rc"
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp gpurun_out/rp_test.err gpurun_out/rp_fail_$it.err; fi
done
echo "failures: $fails"
rm -rf gpurun_out/rp_test
