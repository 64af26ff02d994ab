#!/bin/bash
# quick option sweep of the B=64 reconstruction step (ms_per_step, tapgemm TF/s)
for opts in "" "tg_prefer_nosplit=0" "tg_cfg=0" "tg_cfg=1" "tg_cfg=2" "tg_cfg=0,tg_split=0" "tg_no_split_items=256" "tg_no_split_items=512" "tg_xcd_group=1" "tg_xcd_group=4"; do
  IAN_OPTS="$opts" timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-edit 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('%-40s %.3f ms  %.0f rec/s  tapgemm %.1f TF/s share %.2f' % ('$opts', r['ms_per_step'], r['value'], r['roofline']['achieved'], r['roofline']['tapgemm_share_of_step']))"
done
