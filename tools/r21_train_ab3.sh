#!/bin/bash
# same-box A/B of the training step: VGG target pass batched with the prediction's (no second stream), weight gradients on the main stream
OUT=${1:-gpurun_out/r21_train_ab3.txt}
run() { echo "== $*" >> $OUT; env "$@" RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 4 --no-cpu-baseline --force-allreduce 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith(chr(123)):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['host']['launch_program'])
" >> $OUT 2>&1; }
run SSM_X=0
run SSM_VGG_OVERLAP=0
run SSM_WGRAD_STREAM=0
run SSM_VGG_OVERLAP=0 SSM_WGRAD_STREAM=0
run SSM_X=0
cat $OUT
