#!/bin/bash
# First GPU contact of a round, as ONE gpurun call (run from the repo root on the GPU box):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/first_contact.sh r02'
# Steps are joined with && and a step that times out or dies by a signal stops the sequence (no GPU step after it);
# every step is bounded by its own `timeout -k`; everything lands in gpurun_out/.
set -u
TAG=${1:-r02}
mkdir -p gpurun_out
step() {  # step <name> <seconds> <command...>
    local name=$1 secs=$2; shift 2
    echo "[first_contact] $name"
    timeout -k 10 "$secs" "$@" > "gpurun_out/${TAG}_${name}.log" 2>&1
    local rc=$?
    echo "[first_contact] $name rc=$rc"; tail -3 "gpurun_out/${TAG}_${name}.log"
    # a timeout / kill (124, 137) or a crash by signal (>128) ends the sequence: no further GPU step after it;
    # an ordinary failure (a failing assertion, rc 1) does not
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then return $rc; fi
    return 0
}
step smoke 180 python -c "import __graft_entry__ as g; g.smoke()" &&
step pytest_gpu 600 python -m pytest tests -q -m gpu &&
step bench 300 python bench.py &&
{ grep '^{' "gpurun_out/${TAG}_bench.log" | tail -1 > "gpurun_out/${TAG}_bench.json"; true; } &&
step bench_n400 200 python bench.py --win-ms 25 --no-cpu-baseline &&
step bench_i16 200 python bench.py --sig-dtype i16 --no-cpu-baseline &&
step bench_cfg4_kwta 200 python bench.py --workload cfg4 --kwta exact --steps 200 --no-cpu-baseline &&
step bench_cfg4_kwta_tree 200 python bench.py --workload cfg4 --kwta tree --steps 200 --no-cpu-baseline &&
step ab_n512 200 python tools/ab_bench.py &&
step ab_n512_big 200 python tools/ab_bench.py --batch 4096 --rounds 7 --launches 50 &&
step profile 900 bash tools/profile_bench.sh "$TAG"
echo "[first_contact] done"
