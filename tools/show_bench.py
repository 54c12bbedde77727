import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["input_ests_per_s"], d["ms_per_step"], d["suspensions_per_est"], d["kernel_busy_union_ms"], d["phases_s"])
print(d.get("fresh_batch"))
for o in d.get("other_workloads", []): print({k:v for k,v in o.items() if k not in ("kernels",)})
print(d.get("oneshot")); print(d["roofline"]); print(d.get("cpu_baseline",{}).get("value"))
