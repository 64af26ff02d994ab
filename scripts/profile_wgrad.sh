#!/bin/bash
# Run on the GPU box (via gpurun): backward-weight of every layer in isolation -- event timing, kernel trace, PMC passes.
# usage: scripts/profile_wgrad.sh <tag> [full]      (IAN_OPTS / LAYERS / B pass through)
set -u
TAG=${1:-r06_wgrad}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python scripts/exp/wgrad_probe.py"
OUT=$PWD/$OUT
OUT=$OUT/rates.json $CMD > $OUT/rates.log 2>&1; cat $OUT/rates.log | tail -12
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
if [ "${2:-}" = "full" ]; then
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pmc pass $name failed"
done
fi
python scripts/exp/wgrad_probe_summary.py $OUT $OUT/rates.json $OUT/summary.json | tee $OUT/summary.txt
