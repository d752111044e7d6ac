#!/bin/bash
# full -m gpu tier, then the stream-count sweep of the headline -- usage: gpu_call_full.sh <tag>
TAG=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_gpu_pytest.log
[ $rc -eq 0 ] && bash tools/gpu_call_streams.sh $TAG
