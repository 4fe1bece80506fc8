#!/bin/bash
# Profile a benchmark on the GPU box: kernel trace + stats, then PMC passes (each in its own run, as the guide and
# gpurun require: --pmc never together with a trace domain).
#   tools/profile.sh <tag> [bench.py args...]                       # bench.py (the headline)
#   tools/profile.sh <tag> --only-config c5                         # one of bench.py's BASELINE configurations
#   PROG=tools/bench_configs.py tools/profile.sh <tag> c4 ...        # another program of this repo
# Output: gpurun_out/prof_<tag>/{kt,pmc_*}/...csv, bench.json (the program's stdout of the kernel-trace pass).
# The program itself comes directly after `--` (python3 <script>): no env / bash -c hop under rocprofv3.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
PROG=${PROG:-bench.py}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ "$PROG" = "bench.py" ]; then
  case " $* " in
    *" --only-config "*) KT_ARGS="--config-cpu-seconds 0"; PMC_ARGS="--profile";;   # one BASELINE configuration alone
    *) KT_ARGS="--cpu-sample 0 --configs none"; PMC_ARGS="--cpu-sample 0 --no-two-loop --steps 1 --warmup 0 --configs none";;
  esac
else KT_ARGS="--cpu-seconds 0"; PMC_ARGS="--cpu-seconds 0 --reps 1"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/$PROG" $KT_ARGS "$@" > "$OUT/bench.json" 2> "$OUT/kt.err"
echo "kernel-trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  if [ -n "${PMC_ONLY_TRAFFIC:-}" ] && [ "$C" != FETCH_SIZE ] && [ "$C" != WRITE_SIZE ]; then continue; fi
  D=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$D" -- python3 "$R/$PROG" $PMC_ARGS "$@" > "$OUT/pmc_$D.log" 2>&1
  echo "pmc $C rc=$?"
done
find "$OUT" -name "*.csv" | head -40
