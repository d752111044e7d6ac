#!/bin/bash
# After a `tools/gpu_call.sh TAG profile ...` call: copy the judged artefacts from gpurun_out/ (scratch) into profiles/
# (tracked) under the round's name and rebuild profiles/ROOFLINE.md.
#   bash tools/collect_profiles.sh r07 round3
set -eu
TAG=${1:?gpurun tag}; ROUND=${2:?round name}
cd "$(dirname "$0")/.."
specs=""
for wl in headline n512 cfg4 cfg5 sndenv cfg1 rate48k sndenv_cfg1; do
    [ -f "gpurun_out/${TAG}_${wl}_kernels.json" ] || continue
    for suf in summary.txt kernel_stats.csv kernels.json; do
        cp "gpurun_out/${TAG}_${wl}_${suf}" "profiles/${ROUND}_${wl}_${suf}"
    done
    case $wl in cfg5) specs="$specs ${ROUND}_${wl}=${wl},1280";; *) specs="$specs ${ROUND}_${wl}=${wl}";; esac
done
[ -f gpurun_out/pmc_traffic.json ] && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python3 tools/roofline_table.py $specs > profiles/ROOFLINE.md
echo "profiles/ROOFLINE.md from:$specs"
