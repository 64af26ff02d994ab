#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats + PMC passes for the bench workload.
# usage: scripts/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export IAN_TUNE_CACHE=$PWD/$OUT/tune.txt   # first (un-profiled) run tunes, profiled passes replay its choices
BENCH="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-edit --no-train --no-full-ian --no-bf16x3 $*"
echo "$BENCH   (under rocprofv3 --kernel-trace --stats / --pmc <set>; scripts/profile_round.sh)" > $OUT/cmd.txt
echo "== bench (full) ==" 
if [ -z "${SKIP_FULL_BENCH:-}" ]; then timeout 600 python bench.py $* > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json; fi
echo "== kernel trace =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
ls $OUT/trace | head
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | cut -d' ' -f1)
  echo "== pmc $pmc =="
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pmc pass $name failed"
  ls $OUT/pmc_$name | head -3
done
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)|SQ_|TCC_|GRBM" | head -400 > $OUT/counters.txt
echo done
