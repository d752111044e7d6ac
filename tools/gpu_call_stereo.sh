#!/bin/bash
# cfg5 as interleaved stereo clips (two strided items per clip) -- usage: gpu_call_stereo.sh <tag>
TAG=${1:-r03o}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -m gpu -k "dtypes or stereo or n2048" -p no:cacheprovider > gpurun_out/${TAG}_pytest_stereo.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_pytest_stereo.log
[ $rc -eq 0 ] || exit $rc
for c in f64 f32; do
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --no-cpu-baseline --stereo --compute $c --report-anyway > gpurun_out/${TAG}_bench_cfg5_stereo_$c.json 2> gpurun_out/${TAG}_bench_cfg5_stereo_$c.err; echo "cfg5 stereo $c rc=$?"
done
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --no-cpu-baseline --stereo --sig-dtype i16 > gpurun_out/${TAG}_bench_cfg5_stereo_i16.json 2> gpurun_out/${TAG}_bench_cfg5_stereo_i16.err; echo "cfg5 stereo i16 rc=$?"
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg5_stereo_f64","cfg5_stereo_f32","cfg5_stereo_i16"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], d["roofline"]["frac"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"], d["config"].get("layout"))
PY
