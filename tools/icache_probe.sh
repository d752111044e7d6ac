set -u
ROOT=$(pwd)
mkdir -p gpurun_out/icache
cd /tmp && export TMPDIR=/tmp
for cfg in rate_48k_n1200_nf32 cfg1_44k_n1103_nf32; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $ROOT/gpurun_out/icache/$cfg -- python3 $ROOT/tools/ab_bench.py --cfg $cfg --batch 2048 --launches 10 --rounds 3 --warm 5 > $ROOT/gpurun_out/icache_$cfg.log 2>&1
rc=$?; echo "rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob("$ROOT/gpurun_out/icache/$cfg/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")[:60]; c=r.get("Counter_Name"); v=float(r.get("Counter_Value",0))
        d[k][c][0]+=1; d[k][c][1]+=v
for k,cs in d.items():
    if "melspec" in k:
        print(k, {c: round(t/n,1) for c,(n,t) in cs.items()})
PY
done
