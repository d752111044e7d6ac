#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root.  Three passes of the same bench command:
#   1. rocprofv3 --kernel-trace --stats      -> per-kernel time      (profiles/<tag>_kernel_stats.csv)
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE   (own pass, as MI355X_MICROARCH.md prescribes)
#   3. rocprofv3 --kernel-trace --pmc WRITE_SIZE
# and a one-page summary (profiles/<tag>_summary.txt) via tools/rocprof_summary.py.
# Usage: bash tools/profile_bench.sh r02 [extra bench.py args]
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --only-headline --launch eager --min-seconds 0.02 $*"
export AUD_PROFILE_ARGS="$*"
case " $* " in *" --batch "*) export AUD_PROFILE_BATCH=$(echo " $* " | sed -e 's/.* --batch \([0-9]*\) .*/\1/');; esac
pass() {  # pass <name> <rocprofv3 args...>: a timeout or a death by signal ends the script (no GPU step after it)
    local name=$1; shift
    echo "[profile] $name pass"
    timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
    local rc=$?
    if [ $rc -ne 0 ]; then echo "[profile] $name pass rc=$rc"; fi
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "[profile] stopping after a timeout/crash"; exit $rc; fi
}
pass stats --kernel-trace --stats
pass fetch --kernel-trace --pmc FETCH_SIZE
pass write --kernel-trace --pmc WRITE_SIZE
# optional SQ passes (8 SQ slots per pass; a pass with a counter this ROCm does not know simply fails and is skipped)
pass sqa --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sqb --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
# vector-memory issue / write path (round 5: what bounds the whole-ProcessSegment kernel -- stores or vector issue?)
pass sqc --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CYCLES
# (a TA_* pass -- TA_TA_BUSY_sum, TA_ADDR_STALLED_BY_TC_CYCLES_sum ... -- left a dispatch incomplete and ran into the pass's 300 s
#  limit on this pool, round 5: not collected)
python3 "$ROOT/tools/rocprof_summary.py" "$OUT" "$TAG" > "$ROOT/profiles/${TAG}_summary.txt" 2>&1
cat "$ROOT/profiles/${TAG}_summary.txt"
for f in $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); do cp "$f" "$ROOT/profiles/${TAG}_kernel_stats.csv"; done
# only gpurun_out/ travels back from the GPU box: copy what has to be committed under profiles/
cp "$ROOT/profiles/${TAG}_summary.txt" "$ROOT/profiles/${TAG}_kernel_stats.csv" "$ROOT/profiles/${TAG}_kernels.json" "$ROOT/profiles/pmc_traffic.json" "$ROOT/gpurun_out/" 2>/dev/null
true
