#!/bin/bash
# N = 1103 (prime; BASELINE configs[0] parameters): generic kernel with the Bluestein route -- usage: gpu_call_cfg1.sh <tag>
TAG=${1:-r04c}
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -q -m gpu -k "n1103 or cfg1 or tone or fuzz or generic or stereo or dtypes or golden" -p no:cacheprovider > gpurun_out/${TAG}_pytest_generic.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_pytest_generic.log
[ $rc -eq 0 ] || exit $rc
for c in f64 f32; do
timeout -k 10 300 python bench.py --workload cfg1 --compute $c --no-cpu-baseline --report-anyway > gpurun_out/${TAG}_bench_cfg1_$c.json 2> gpurun_out/${TAG}_bench_cfg1_$c.err; echo "cfg1 $c rc=$?"; tail -c 200 gpurun_out/${TAG}_bench_cfg1_$c.err
done
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg1_f64","cfg1_f32"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"], d["config"]["streams"])
PY
