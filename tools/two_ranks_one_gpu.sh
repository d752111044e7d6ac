#!/bin/bash
# Run ON THE GPU BOX: the N > 1 flow of bench.py as TWO processes on the ONE GPU of the box (gloo as the control plane, the
# library's direct device-to-device reassembly as the collective -- RCCL refuses two ranks on one device).  Timing means
# nothing here (the ranks share the chip); what it checks on real hardware is the two-rank control flow: shards, the
# inter-process buffer export, the graph-captured pushes, rank 0's parity check of BOTH ranks' blocks.
#   bash tools/two_ranks_one_gpu.sh TAG [RANKS]      (RANKS 2..4: with the caller's own process that stays inside the box's
#                                                      limit of 6 processes on the card; 3 and 4 exercise one flag per PEER)
set -u
TAG=${1:-two}
N=${2:-2}
MODE=${3:-direct}     # direct | host
if [ "$N" -lt 2 ] || [ "$N" -gt 4 ]; then echo "RANKS must be 2..4"; exit 2; fi
mkdir -p gpurun_out
PORT=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((N - 1))); do
    RANK=$r LOCAL_RANK=0 WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
        timeout -k 10 400 python bench.py --gpus "$N" --dist-backend gloo --gather-mode "$MODE" --no-cpu-baseline --no-stream-read \
        --steps 20 --warmup 5 --min-seconds 0.05 --cfg3-total 1024 > "gpurun_out/${TAG}_rank$r.log" 2>&1 &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
echo "[two_ranks] rc=$rc"
tail -n 3 "gpurun_out/${TAG}_rank1.log" | cut -c1-300
grep '^{' "gpurun_out/${TAG}_rank0.log" | tail -n 1 > "gpurun_out/${TAG}_line.json"
python3 - "$TAG" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/%s_line.json" % sys.argv[1]))
    print("n_gpus", d["n_gpus"], "scaling", d["scaling"], "ranks", d.get("rccl_ranks"), "gathered", d.get("gathered_shape"))
    print("parity", d["parity"]["pass"], d["parity"]["max_scaled_err"], d["parity"]["checked"])
    print("collective", d["collective"][:100])
    print("arrival_timeouts", d.get("arrival_timeouts"), "direct_gather", json.dumps(d.get("direct_gather"))[:300])
except Exception as ex:
    print("no JSON line:", ex)
PY
exit $rc
