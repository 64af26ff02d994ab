#!/bin/bash
# Timing-only ablations of the tapgemm K loop (IAN_simple batch 64, every layer forced to 64x64 tiles, no autotune).
# usage (on the GPU box): scripts/ablate_tapgemm.sh <outdir>
# The ablation variants (10/11/12) produce WRONG results by design; they only exist in a library built with -DIAN_ABLATION.
OUT=${1:-gpurun_out/ablate}; mkdir -p $OUT
export IAN_ABLATION_BUILD=1
for v in 2 3 10 11 12 1 0; do
  IAN_NO_AUTOTUNE=1 IAN_OPTS=tg_cfg=2,tg_variant=$v timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-edit --no-train --no-full-ian 2>/dev/null \
    | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('variant $v  ms/step %.4f  tapgemm avg launch ms %.4f  frac %.3f' % (r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"
done | tee $OUT/ablate.txt
