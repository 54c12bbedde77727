#!/bin/bash
# GPU box: HBM traffic of every kernel of the bench from the TCC counters, two separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing beside --pmc).  A smaller batch keeps
# the serialised counter runs short; traffic is reported per launch.
#   bash tools/pmc_traffic.sh [ESTS [WORKLOAD]]   -> gpurun_out/pmc_traffic[_<workload>].json
set -e
export TMPDIR=/tmp
N=${1:-20000}
W=${2:-C3}
export PMC_ESTS=$N
export PMC_TAG=$([ "$W" = C3 ] && echo "" || echo "_$(echo $W | tr 'A-Z' 'a-z')")
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_$C -o pmc -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-oneshot --ests $N > gpurun_out/pmc_$C.json 2> gpurun_out/pmc_$C.err || { tail -5 gpurun_out/pmc_$C.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"]
            out[k][c] += float(row["Counter_Value"])
            if c == "FETCH_SIZE":
                out[k]["launches"] += 1
res = {}
for k, v in out.items():
    n = max(v["launches"], 1)
    # FETCH_SIZE / WRITE_SIZE are KiB; gfx950 tallies 128-B read requests at 64 B -> double the reads
    res[k] = {"launches": v["launches"], "fetch_KiB": v["FETCH_SIZE"], "write_KiB": v["WRITE_SIZE"],
              "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / n}
import os
res["_run"] = {"ests": int(os.environ.get("PMC_ESTS", "20000")), "note": "bench.py --steps 1 --warmup 0 --ests <ests>: the launches of this run carry a different number of ESTs than the full bench's"}
json.dump(res, open("gpurun_out/pmc_traffic%s.json" % os.environ.get("PMC_TAG", ""), "w"), indent=1, sort_keys=True)
res.pop("_run")
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["launches"])[:12]:
    print(k[:70], v["launches"], round(v["hbm_bytes_per_launch"]))
PY
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name '*.csv' -size +2M -delete
