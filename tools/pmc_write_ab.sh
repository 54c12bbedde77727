#!/bin/bash
# GPU box: WRITE_SIZE of dp_batch_kernel over one 20 000-EST C3 step under a few settings (one --pmc pass each).
#   bash tools/pmc_write_ab.sh "A=1" "PGPU_ALIGN_BAND=0" ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "$@"; do
  ( for kv in $(echo $cfg | tr ',' ' '); do export $kv; done
    rm -rf gpurun_out/pmc_ab
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_ab -o pmc -- python3 bench.py --workload C3 --steps 1 --warmup 0 --no-cpu --no-oneshot --ests 20000 > gpurun_out/pmc_ab.json 2> gpurun_out/pmc_ab.err || { tail -3 gpurun_out/pmc_ab.err; exit 1; }
    python3 - "$cfg" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_ab/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") == "WRITE_SIZE":
            k = row["Kernel_Name"][:50]; tot[k] += float(row["Counter_Value"]); n[k] += 1
print("== %s" % sys.argv[1])
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:int(__import__("os").environ.get("PMC_TOP", "4"))]:
    print("   %-52s %6d launches  %10.1f MiB written" % (k, n[k], v / 1024.0))
PY
  )
done
