# Run ON THE GPU BOX: sweeps the Bluestein transform length of the generic kernel for N = 1103 (bench.py --workload cfg1) on a
# tuning build (python -m auditory_amd.build --tag tune -DAUD_TUNE_BLUESTEIN=1: AUD_BLUESTEIN_L overrides the plan's choice).
set -u
mkdir -p gpurun_out
for L in 2304 2560 2592 2880 3072 3456 4096 2400; do
  AUD_BLUESTEIN_L=$L timeout -k 10 200 python tools/run_with_lib.py tune bench.py --workload cfg1 --no-cpu-baseline --no-stream-read --only-headline > gpurun_out/tune_cfg1_L$L.log 2>&1
  rc=$?; echo "L=$L rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi
  grep '^{' gpurun_out/tune_cfg1_L$L.log | tail -1 > gpurun_out/tune_cfg1_L$L.json
  python3 -c "
import json
d=json.load(open('gpurun_out/tune_cfg1_L$L.json'))
print('L=$L', 'us per 256 segments', d['us_per_step_device']['mean'], 'alone', d['roofline']['avg_launch_us'], 'parity', d['parity']['pass'], d['parity']['max_scaled_err'])"
done
