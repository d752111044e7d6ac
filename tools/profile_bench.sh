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
BENCH="python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --launch eager $*"
echo "[profile] stats pass";  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1 || echo "stats pass rc=$?"
echo "[profile] FETCH_SIZE pass"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1 || echo "fetch pass rc=$?"
echo "[profile] WRITE_SIZE pass"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1 || echo "write pass rc=$?"
python3 "$ROOT/tools/rocprof_summary.py" "$OUT" "$TAG" > "$ROOT/profiles/${TAG}_summary.txt" 2>&1
cat "$ROOT/profiles/${TAG}_summary.txt"
for f in $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); do cp "$f" "$ROOT/profiles/${TAG}_kernel_stats.csv"; done
cp "$ROOT/profiles/${TAG}_summary.txt" "$ROOT/profiles/${TAG}_kernel_stats.csv" "$ROOT/gpurun_out/" 2>/dev/null
