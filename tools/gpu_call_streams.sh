#!/bin/bash
TAG=${1:-r02q}
mkdir -p gpurun_out
for st in 1 2 3 4; do
  timeout -k 10 200 python bench.py --only-headline --no-cpu-baseline --streams $st > gpurun_out/${TAG}_bench_streams$st.json 2> gpurun_out/${TAG}_bench_streams$st.err; echo "streams $st rc=$?"
done
timeout -k 10 200 python bench.py --only-headline --no-cpu-baseline --streams 2 --compute f32 --report-anyway > gpurun_out/${TAG}_bench_streams2_f32.json 2> gpurun_out/${TAG}_bench_streams2_f32.err
python - "$TAG" <<'PY'
import json,sys
for f in ("streams1","streams2","streams3","streams4","streams2_f32"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"].get("pipelined_GBps"), d["parity"]["n_past_1e-5"])
PY
