#!/bin/bash
# same-box A/B of the training step (BASELINE config 3) with the Winograd-domain weight gradients on / off and their workgroups-per-launch target
OUT=${1:-gpurun_out/r17_train_ab.txt}
run() { echo "== $*" >> $OUT; env "$@" python bench.py --mode train --precision f32w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d.get('time_split_ms_per_step'), d.get('roofline',{}).get('frac'))
" >> $OUT 2>&1; }
run SSM_WGRAD_WINO=0
run SSM_WGRAD_WINO=1 SSM_WGRADW_TARGET=64
run SSM_WGRAD_WINO=1 SSM_WGRADW_TARGET=128
run SSM_WGRAD_WINO=1 SSM_WGRADW_TARGET=256
run SSM_WGRAD_WINO=0
cat $OUT
