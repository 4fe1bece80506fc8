#!/bin/bash
# PMC counters of My_dsyev's kernels (one rocprofv3 --pmc pass per counter group, no trace domain): tools/dsyev_pmc.sh <N|V> <n> <tag>
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
JOB="$1"; N="$2"; TAG="$3"
OUT="$R/gpurun_out/pmc_dsyev_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- python3 "$R/tools/dsyev_once.py" "$JOB" "$N" > "$OUT/p$i.log" 2>&1
  echo "pmc $C rc=$?"
done
python3 "$R/tools/dsyev_pmc_summary.py" "$OUT"
