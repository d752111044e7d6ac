#!/bin/bash
# VGPRs / scratch / occupancy of the wave kernels (static, no GPU):  bash tools/wave_resources.sh [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && for f in melspec_w16 melspec_w20 melspec_w64; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I"$ROOT/include" \
    -I"$ROOT/auditory_amd/csrc" -c "$ROOT/auditory_amd/csrc/$f.hip" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1; done | python3 -c '
import re, sys, subprocess
row = {}
for l in sys.stdin:
    m = re.search(r"remark: [^ ]* *(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", l)
    if not m:
        continue
    if m.group(1) == "Function Name":
        row = {}
    row[m.group(1)] = m.group(2)
    if m.group(1).startswith("LDS Size"):
        name = subprocess.run(["c++filt", row["Function Name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void aud::\(anonymous namespace\)::", "", name).split(">(")[0] + ">"
        print("%-48s VGPRs %4s  SGPRs %4s  scratch %4s  waves/SIMD %s" % (name, row["VGPRs"], row["TotalSGPRs"], row["ScratchSize [bytes/lane]"], row["Occupancy [waves/SIMD]"]))
'
