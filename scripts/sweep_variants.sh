#!/bin/bash
# within-box A/B of the tapgemm K-loop schedules on the bench workload (two rounds, interleaved)
for round in 1 2; do
for opts in "tg_variant=0" "tg_variant=1" "tg_variant=2"; do
  IAN_OPTS="$opts" timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-edit --no-train 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('%-24s %.3f ms  %.0f rec/s  tapgemm %.1f TF/s (frac %.3f) share %.2f' % ('$opts', r['ms_per_step'], r['value'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['tapgemm_share_of_step']))"
done; done
