set -e
export TMPDIR=/tmp
rm -rf gpurun_out/trace1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace1 -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu --ests 60000 > gpurun_out/trace1.json 2> gpurun_out/trace1.err || { tail -5 gpurun_out/trace1.err; exit 1; }
ls -la gpurun_out/trace1/*
