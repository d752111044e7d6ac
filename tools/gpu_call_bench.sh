#!/bin/bash
# the bench line as the driver runs it, then with the default flags -- usage: gpu_call_bench.sh <tag>
TAG=${1:-r02h}
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err; echo "driver-style rc=$?"; tail -c 600 gpurun_out/${TAG}_bench_driver.err
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "default rc=$?"; tail -c 300 gpurun_out/${TAG}_bench_default.err
python - "$TAG" <<'PY'
import json,sys
for f in ("driver","default"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1] if len(sys.argv)>1 else "r02h", f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["ms_per_step"], d["us_per_step_device"], d["roofline"]["frac"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"])
    for k,v in (d.get("modes") or {}).items(): print("  mode", k, v["value"], v["us_per_step_device"]["median"], (v.get("parity") or {}).get("max_scaled_err"), (v.get("parity") or {}).get("n_past_1e-5"))
    for k,v in (d.get("also") or {}).items(): print("  also", k, v["value"], v["us_per_step_device"]["median"], (v.get("parity") or {}).get("max_scaled_err"), (v.get("parity") or {}).get("n_past_1e-5"))
    print("  cfg3", d.get("cfg3")); print("  cpu", d.get("cpu_baseline"))
PY
