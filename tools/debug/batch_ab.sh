for cfg in "--batch 4" "--batch 1"; do
  if [[ "$cfg" == *TWO* ]]; then export CSAM_GROUP_TWO_STREAMS=1; cfg="--batch 4"; else unset CSAM_GROUP_TWO_STREAMS; fi
  python bench.py --steps 20 --warmup 5 $cfg --no-cpu-baseline --no-kernel-timer 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$cfg two=${CSAM_GROUP_TWO_STREAMS:-0}', 'ms/step',round(d['ms_per_step'],2),'kept',c['kept_masks_per_image'],'serial',round(c['serial_leg']['ms_per_step'],2), 'steps med/max', c['per_rank_step_ms_median_max'])"
done
