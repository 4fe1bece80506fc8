#!/bin/bash
# per-kernel times of My_dsyev (rocprofv3 --kernel-trace --stats) on the GPU box: tools/dsyev_profile.sh <N|V> <n> <tag>
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
JOB="$1"; N="$2"; TAG="$3"
OUT="$R/gpurun_out/prof_dsyev_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/tools/dsyev_once.py" "$JOB" "$N" > "$OUT/run.txt" 2> "$OUT/kt.err"
grep -v "^[WE]2026" "$OUT/run.txt"
F=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" "$R/gpurun_out/dsyev_${TAG}_kernel_stats.csv"; cut -c1-160 "$F" | head -24; else echo "no kernel_stats.csv"; fi
