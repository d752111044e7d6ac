#!/bin/bash
# full -m gpu tier, smoke(), then the bench lines -- usage: gpu_call_final.sh <tag>
TAG=${1:-r03k}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_gpu_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" > gpurun_out/${TAG}_smoke.log 2>&1; rc=$?; tail -2 gpurun_out/${TAG}_smoke.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_call_bench.sh $TAG
