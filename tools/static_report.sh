#!/bin/bash
# Static per-kernel resource report of every HIP source (no GPU needed):
#   bash tools/static_report.sh r01   ->  profiles/r01_static_kernel_resources.txt, profiles/r01_static_isa_mix_r16.txt
set -eu
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT="$ROOT/profiles/${TAG}_static_kernel_resources.txt"
TMP=$(mktemp -d)
echo "# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, -O3 -fno-slp-vectorize (static; no GPU run)" > "$OUT"
for f in "$ROOT"/auditory_amd/csrc/*.hip; do
    [ "$(basename "$f")" = capi.hip ] && continue
    (cd "$TMP" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I"$ROOT/include" \
        -I"$ROOT/auditory_amd/csrc" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1) |
    python3 -c '
import re, sys
row = {}
for l in sys.stdin:
    m = re.search(r"remark: [^ ]* *(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", l)
    if not m:
        continue
    if m.group(1) == "Function Name":
        row = {}
    row[m.group(1)] = m.group(2)
    if m.group(1).startswith("LDS Size"):
        print("\t".join("%s: %s" % kv for kv in row.items()))
' >> "$OUT"
done
(cd "$TMP" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I"$ROOT/include" \
    -I"$ROOT/auditory_amd/csrc" -S --cuda-device-only -o r16.s "$ROOT/auditory_amd/csrc/melspec_r16.hip" 2>/dev/null)
{
    echo "# tools/isa_mix.py on k_melspec_r16<float, DIRECT, 1 tile> (the bench kernel), static counts over all sample routes"
    python3 "$ROOT/tools/isa_mix.py" "$TMP/r16.s" IfLb1ELi1
    echo "# k_melspec_r16<float, DIRECT, 2 tiles>"
    python3 "$ROOT/tools/isa_mix.py" "$TMP/r16.s" IfLb1ELi2
} > "$ROOT/profiles/${TAG}_static_isa_mix_r16.txt"
rm -rf "$TMP"
echo "wrote $OUT"
