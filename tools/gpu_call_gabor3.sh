#!/bin/bash
TAG=${1:-r03r}
mkdir -p gpurun_out
for st in 2 3 4; do
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline --streams $st > gpurun_out/${TAG}_bench_cfg4_f64_s$st.json 2> gpurun_out/${TAG}_bench_cfg4_f64_s$st.err; echo "cfg4 f64 streams $st rc=$?"
done
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline --streams 3 --option gabor_lds=2 > gpurun_out/${TAG}_bench_cfg4_f64_s3g2.json 2> gpurun_out/${TAG}_bench_cfg4_f64_s3g2.err
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg4_f64_s2","cfg4_f64_s3","cfg4_f64_s4","cfg4_f64_s3g2"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], d["config"]["kernel"], d["parity"]["n_past_1e-5"])
PY
