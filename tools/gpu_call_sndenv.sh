#!/bin/bash
# the whole SndEnv.ProcessSegment loop (mel + Power + LogPower + MFCC tail) -- usage: gpu_call_sndenv.sh <tag>
TAG=${1:-r03t}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -m gpu -k "mfcc or sndenv or smooth" -p no:cacheprovider > gpurun_out/${TAG}_pytest_mfcc.log 2>&1; rc=$?; tail -2 gpurun_out/${TAG}_pytest_mfcc.log
[ $rc -eq 0 ] || exit $rc
for c in f64 f32; do
timeout -k 10 300 python bench.py --workload sndenv --no-cpu-baseline --compute $c --report-anyway > gpurun_out/${TAG}_bench_sndenv_$c.json 2> gpurun_out/${TAG}_bench_sndenv_$c.err; echo "sndenv $c rc=$?"; tail -c 300 gpurun_out/${TAG}_bench_sndenv_$c.err
done
python - "$TAG" <<'PY'
import json,sys
for f in ("sndenv_f64","sndenv_f32"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["algorithmic_bytes_per_launch"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"], d["config"]["streams"])
PY
bash tools/gpu_call_prof_sndenv.sh $TAG
