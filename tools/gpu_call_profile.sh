#!/bin/bash
# rocprofv3 evidence for the headline (N = 400, float64) and the neighbouring modes -- usage: gpu_call_profile.sh <tag>
TAG=${1:-r02}
bash tools/profile_bench.sh ${TAG}_n400_f64 > gpurun_out/${TAG}_prof_n400_f64.log 2>&1; tail -4 gpurun_out/${TAG}_prof_n400_f64.log
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_headline.json
bash tools/profile_bench.sh ${TAG}_n400_f32 --compute f32 > gpurun_out/${TAG}_prof_n400_f32.log 2>&1; tail -2 gpurun_out/${TAG}_prof_n400_f32.log
AUD_PROFILE_BATCH=4096 bash tools/profile_bench.sh ${TAG}_n400_f64_b4096 --batch 4096 --ring-mb 600 --steps 40 > gpurun_out/${TAG}_prof_n400_f64_b4096.log 2>&1; tail -2 gpurun_out/${TAG}_prof_n400_f64_b4096.log
cp gpurun_out/${TAG}_pmc_traffic_headline.json gpurun_out/pmc_traffic.json
