#!/bin/bash
# Developer: the headline loop at several group sizes on ONE box (boxes differ by a few per cent).  bash tools/debug/batch_ab.sh [steps] [sizes..]
steps=${1:-40}; shift
for b in ${@:-4 1}; do
  python bench.py --steps $steps --warmup 5 --batch $b --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('--batch $b', 'ms/step',round(d['ms_per_step'],2),'kept',c['kept_masks_per_image'],'serial',round(c['serial_leg']['ms_per_step'],2), 'steps med/max', c['per_rank_step_ms_median_max'])"
done
