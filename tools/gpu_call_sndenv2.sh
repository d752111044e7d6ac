#!/bin/bash
TAG=${1:-r03z}
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload sndenv --no-cpu-baseline --option n400_geometry=25 > gpurun_out/${TAG}_bench_sndenv_f64_w25.json 2> gpurun_out/${TAG}_bench_sndenv_f64_w25.err; echo "rc=$?"
timeout -k 10 300 python bench.py --workload sndenv --no-cpu-baseline --streams 3 > gpurun_out/${TAG}_bench_sndenv_f64_s3.json 2> gpurun_out/${TAG}_bench_sndenv_f64_s3.err; echo "rc=$?"
timeout -k 10 300 python bench.py --workload sndenv --no-cpu-baseline --streams 6 > gpurun_out/${TAG}_bench_sndenv_f64_s6.json 2> gpurun_out/${TAG}_bench_sndenv_f64_s6.err; echo "rc=$?"
python - "$TAG" <<'PY'
import json,sys
for f in ("sndenv_f64_w25","sndenv_f64_s3","sndenv_f64_s6"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], d["roofline"]["avg_launch_us"], d["config"]["kernel"], d["parity"]["n_past_1e-5"], d["config"]["streams"])
PY
