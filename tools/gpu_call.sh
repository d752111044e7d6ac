#!/bin/bash
# One gpurun call = a sequence of bounded steps (run from the repo root on the GPU box):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/gpu_call.sh TAG step [step ...]'
# Every step is bounded by its own `timeout -k`, writes gpurun_out/TAG_<step>.log, and a step that times out or dies by a
# signal ends the sequence (no GPU step after it); an ordinary failure (rc 1) does not.
# Steps: smoke | pytest | pytest_all | bench | bench:<extra bench.py args joined by ','> | ab | profile[:<bench args>] | stamps | mfma |
#        stream | wstream | dispatch | ubench | two_ranks | ranks:<2..4> | hostranks:<2..4> | hostcall | tune_bluestein |
#        sweep[:<rate_sweep.py args>] |
#        runlib:<tag>,<script.py>,<args...>  (the script against an experimental build: python -m auditory_amd.build --tag <tag> -D...)
set -u
TAG=${1:?tag}; shift
mkdir -p gpurun_out
run() {  # run <name> <seconds> <command...>
    local name=$1 secs=$2; shift 2
    echo "[gpu_call] $name"
    timeout -k 10 "$secs" "$@" > "gpurun_out/${TAG}_${name}.log" 2>&1
    local rc=$?
    echo "[gpu_call] $name rc=$rc"; tail -4 "gpurun_out/${TAG}_${name}.log" | cut -c1-600
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "[gpu_call] stopping after a timeout/crash"; exit $rc; fi
    return 0
}
i=0
for step in "$@"; do
    i=$((i + 1))
    case "$step" in
        smoke) run smoke 240 python -c "import __graft_entry__ as g; g.smoke()" ;;
        pytest) run pytest 900 python -m pytest tests -q -m gpu -x ;;
        pytest_all) run pytest 900 python -m pytest tests -q -m gpu ;;
        bench) run bench$i 400 python bench.py
               grep '^{' "gpurun_out/${TAG}_bench$i.log" | tail -1 > "gpurun_out/${TAG}_bench$i.json" ;;
        bench:*) args=$(echo "${step#bench:}" | tr ',' ' ')
               run bench$i 400 python bench.py $args
               grep '^{' "gpurun_out/${TAG}_bench$i.log" | tail -1 > "gpurun_out/${TAG}_bench$i.json" ;;
        ab) run ab$i 300 python tools/ab_bench.py ;;
        ab:*) args=$(echo "${step#ab:}" | tr ',' ' ')
               run ab$i 300 python tools/ab_bench.py $args ;;
        profile) run profile_headline 900 bash tools/profile_bench.sh "${TAG}_headline" ;;
        profile:*) args=$(echo "${step#profile:}" | tr ',' ' ')   # tag = TAG_<workload>
               wl=$(echo " $args " | sed -n -e 's/.* --workload \([a-z0-9_]*\) .*/\1/p'); wl=${wl:-headline}
               run profile_$wl 900 bash tools/profile_bench.sh "${TAG}_$wl" $args ;;
        stamps:*) args=$(echo "${step#stamps:}" | tr ',' ' ')
               run stamps$i 400 python tools/stamp_profile.py $args ;;
        trace) mkdir -p "gpurun_out/trace_${TAG}"
               (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/trace_${TAG}" -- \
                    python3 "$OLDPWD/bench.py" --only-headline --no-cpu-baseline --min-seconds 0.05 > "$OLDPWD/gpurun_out/${TAG}_trace.log" 2>&1)
               rc=$?; echo "[gpu_call] trace rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi
               python3 tools/trace_overlap.py "gpurun_out/trace_${TAG}" > "gpurun_out/${TAG}_graph_overlap.txt" 2>&1; cat "gpurun_out/${TAG}_graph_overlap.txt" ;;
        ubench) (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates valu_rates.hip) &&
                run ubench 120 /tmp/valu_rates ;;
        mfma) (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_beside_valu mfma_beside_valu.hip) &&
                run mfma 120 /tmp/mfma_beside_valu ;;
        stream) (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DSTREAM_READ_MAIN -o /tmp/stream_read stream_read.hip) &&
                run stream 120 /tmp/stream_read ;;
        runlib:*) args=$(echo "${step#runlib:}" | tr ',' ' ')   # runlib:TAG,script.py,args...  (an experimental build, tools/run_with_lib.py)
               run runlib$i 400 python tools/run_with_lib.py $args
               grep '^{' "gpurun_out/${TAG}_runlib$i.log" | tail -1 > "gpurun_out/${TAG}_runlib$i.json" ;;
        tune_bluestein) run tune_bluestein 1000 bash tools/tune_bluestein.sh ;;
        wstream) (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_write stream_write.hip) &&
                run wstream 120 /tmp/stream_write ;;
        two_ranks) run two_ranks 500 bash tools/two_ranks_one_gpu.sh "${TAG}_two" ;;
        ranks:*) run ranks${step#ranks:} 500 bash tools/two_ranks_one_gpu.sh "${TAG}_ranks${step#ranks:}" "${step#ranks:}" ;;
        hostranks:*) run hostranks${step#hostranks:} 500 bash tools/two_ranks_one_gpu.sh "${TAG}_hostranks${step#hostranks:}" "${step#hostranks:}" host ;;
        hostcall) run hostcall 300 python tools/host_call_time.py ;;
        sweep) run sweep 600 python tools/rate_sweep.py --json "gpurun_out/${TAG}_rate_sweep.json" ;;
        sweep:*) args=$(echo "${step#sweep:}" | tr ',' ' ')
               run sweep$i 600 python tools/rate_sweep.py --json "gpurun_out/${TAG}_rate_sweep$i.json" $args ;;
        dispatch) (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_rate dispatch_rate.hip) &&
                run dispatch 120 /tmp/dispatch_rate ;;
        *) echo "unknown step $step"; exit 2 ;;
    esac
done
echo "[gpu_call] done"
