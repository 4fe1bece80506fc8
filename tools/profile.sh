#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run).
# usage: tools/profile.sh <tag> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --cpu-sample 0 "$@" > "$OUT/kt.log" 2>&1
echo "kernel-trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  D=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$D" -- python3 "$R/bench.py" --cpu-sample 0 --no-two-loop --steps 1 --warmup 0 "$@" > "$OUT/pmc_$D.log" 2>&1
  echo "pmc $C rc=$?"
done
find "$OUT" -name "*.csv" | head -40
