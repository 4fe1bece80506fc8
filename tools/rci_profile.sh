#!/bin/bash
# per-kernel times of the reverse-communication path (rocprofv3 --kernel-trace --stats): tools/rci_profile.sh <batch> <mode> <tag>
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
B="$1"; MODE="$2"; TAG="$3"
OUT="$R/gpurun_out/prof_rci_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/tools/bench_configs.py" rci --rci-batch "$B" --rci-modes "$MODE" --cpu-seconds 0 > "$OUT/run.txt" 2> "$OUT/kt.err"
grep -v "^[WE]2026" "$OUT/run.txt" | cut -c1-400
F=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" "$R/gpurun_out/rci_${TAG}_kernel_stats.csv"; cut -c1-150 "$F" | head -16; else echo "no kernel_stats.csv"; fi
