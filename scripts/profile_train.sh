#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE PMC passes (separate, as the guide prescribes)
# of 4 updates of the full-IAN training step at 128 images per GPU.   usage: scripts/profile_train.sh <tag>
set -u
TAG=${1:-r04_train}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export IAN_TUNE_CACHE=$PWD/$OUT/tune.txt   # the first pass tunes, the PMC passes replay its choices
CMD="python scripts/train_profile.py"
echo "B=128 ITERS=4 $CMD   (under rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE; scripts/profile_train.sh)" > $OUT/cmd.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
ls $OUT/trace | head -4
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$pmc -o pmc -- $CMD > $OUT/pmc_$pmc.log 2>&1 || echo "pmc pass $pmc failed"
  ls $OUT/pmc_$pmc | head -3
done
echo done
