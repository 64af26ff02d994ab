#!/bin/bash
# Run on the GPU box (via gpurun): EVERY committed profile of a round in one call -- batch-64 headline, full IAN batch 256, the training
# step, the batch-1 chains -- summarised ON the box (the raw rocprofv3 CSVs of the four together exceed what gpurun copies back) into
# gpurun_out/<tag>/: copy that directory's files into profiles/.        usage: scripts/profile_all.sh <tag, e.g. r06>
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash scripts/profile_round.sh ${TAG}_raw_b64 2>&1 | tail -2
python scripts/summarize_profile.py gpurun_out/${TAG}_raw_b64 $OUT/${TAG}_ian_simple_b64 | tail -1
cp gpurun_out/${TAG}_raw_b64/bench.json $OUT/${TAG}_bench_line.json
SKIP_FULL_BENCH=1 bash scripts/profile_round.sh ${TAG}_raw_b256 --arch IAN 2>&1 | tail -2
python scripts/summarize_profile.py gpurun_out/${TAG}_raw_b256 $OUT/${TAG}_ian_b256 | tail -1
bash scripts/profile_train.sh ${TAG}_raw_train 2>&1 | tail -2
python scripts/summarize_train_profile.py gpurun_out/${TAG}_raw_train/trace/trace_kernel_trace.csv $OUT/${TAG}_train_ian_b128 gpurun_out/${TAG}_raw_b64/bench.json | tail -1
python scripts/train_timeline.py gpurun_out/${TAG}_raw_train/trace/trace_kernel_trace.csv $OUT/${TAG}_train_timeline_after.json > /dev/null
bash scripts/profile_b1.sh ${TAG}_raw_b1 2>&1 | tail -2
python scripts/summarize_b1_profile.py gpurun_out/${TAG}_raw_b1 $OUT/${TAG}_batch1_chains | tail -1
for d in b64 b256 train b1; do tail -c 2000 gpurun_out/${TAG}_raw_$d/*.err gpurun_out/${TAG}_raw_$d/*.log 2>/dev/null | grep -i "error\|fail\|Traceback" | head -5; done
rm -rf gpurun_out/${TAG}_raw_b64 gpurun_out/${TAG}_raw_b256 gpurun_out/${TAG}_raw_train gpurun_out/${TAG}_raw_b1
ls -la $OUT
