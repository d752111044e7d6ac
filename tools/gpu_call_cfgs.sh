#!/bin/bash
# secondary BASELINE rows: cfg4 (gabor), cfg5 (44.1 kHz / N = 2048 / 128 mel, >= 1 GB resident), cfg1 parameters (N = 1103)
TAG=${1:-r02k}
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests -q -m gpu -k "gabor or stereo" -p no:cacheprovider > gpurun_out/${TAG}_pytest_gabor.log 2>&1; tail -2 gpurun_out/${TAG}_pytest_gabor.log
timeout -k 10 300 python bench.py --workload cfg1 > gpurun_out/${TAG}_bench_cfg1_f64.json 2> gpurun_out/${TAG}_bench_cfg1_f64.err; echo "cfg1 f64 rc=$?"
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline --option gabor_lds=0 > gpurun_out/${TAG}_bench_cfg4_f64_globalgabor.json 2> gpurun_out/${TAG}_bench_cfg4_f64_globalgabor.err; echo "cfg4 f64 (global-memory gabor) rc=$?"
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg4_f64.json 2> gpurun_out/${TAG}_bench_cfg4_f64.err; echo "cfg4 f64 rc=$?"
timeout -k 10 300 python bench.py --workload cfg4 --compute f32 --no-cpu-baseline --report-anyway > gpurun_out/${TAG}_bench_cfg4_f32.json 2> gpurun_out/${TAG}_bench_cfg4_f32.err; echo "cfg4 f32 rc=$?"
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_cfg5_f64.json 2> gpurun_out/${TAG}_bench_cfg5_f64.err; echo "cfg5 f64 rc=$?"; tail -c 300 gpurun_out/${TAG}_bench_cfg5_f64.err
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --compute f32 --no-cpu-baseline --report-anyway > gpurun_out/${TAG}_bench_cfg5_f32.json 2> gpurun_out/${TAG}_bench_cfg5_f32.err; echo "cfg5 f32 rc=$?"
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg1_f64","cfg4_f64_globalgabor","cfg4_f64","cfg4_f32","cfg5_f64","cfg5_f32"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"], d["roofline"]["frac"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"], d.get("cpu_baseline",{}).get("value"))
PY
