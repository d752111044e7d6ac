set -u
mkdir -p gpurun_out
for L in 2304 2400 2500 2560 3072 4096; do
  AUD_BLUESTEIN_L=$L timeout -k 10 200 python tools/run_with_lib.py tune bench.py --workload cfg1 --no-cpu-baseline --no-stream-read --only-headline > gpurun_out/r4g_cfg1_L$L.log 2>&1
  rc=$?; echo "L=$L rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi
  grep '^{' gpurun_out/r4g_cfg1_L$L.log | tail -1 > gpurun_out/r4g_cfg1_L$L.json
done
