#!/bin/bash
# same-box A/B of the training step (BASELINE config 3, RCCL buckets forced at world 1): which layers take the Winograd-domain weight
# gradient, workgroups per launch, the early data-gradient filter repack
OUT=${1:-gpurun_out/r17_train_ab2.txt}
run() { echo "== $*" >> $OUT; env "$@" RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 4 --no-cpu-baseline --force-allreduce 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('samples/s', d['value'], 'ms/step', d['ms_per_step'], 'host enqueue ms', d.get('host_enqueue_ms_per_step'), 'in region', d['host']['enqueue_ms_per_step_in_timed_region'], 'program', d['host']['launch_program'], 'frac', d.get('roofline',{}).get('frac'))
" >> $OUT 2>&1; }
run SSM_X=0
run SSM_DGRAD_PACK_EARLY=0
run SSM_WGRAD_WINO=all
run SSM_WGRAD_WINO=all SSM_WGRADW_TARGET=96
run SSM_WGRAD_WINO=all SSM_WGRADW_TARGET=160
run SSM_WGRADW_TARGET=96
run SSM_X=0
cat $OUT
