#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + FETCH_SIZE / WRITE_SIZE PMC passes (separate, as the guide prescribes) of the
# batch-1 chains (scripts/b1_chain_profile.py).   usage: scripts/profile_b1.sh <tag>
set -u
TAG=${1:-r05_b1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export IAN_TUNE_CACHE=$PWD/$OUT/tune.txt
CMD="python scripts/b1_chain_profile.py"
echo "REPS=100 $CMD   (under rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE; scripts/profile_b1.sh)" > $OUT/cmd.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$pmc -o pmc -- $CMD > $OUT/pmc_$pmc.log 2>&1 || echo "pmc pass $pmc failed"
done
ls $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE | head -12
echo done
