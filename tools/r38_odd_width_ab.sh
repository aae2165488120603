#!/bin/bash
# same-box A/B of the training step: the 11x11 bottleneck layers (conv6.0 / conv6.1, odd width) in the direct form + split-K (r5) vs F(2x2,3x3) + split-K (r6)
OUT=${1:-gpurun_out/r38_odd_width_ab.txt}
run() { echo "== $*" >> $OUT; env "$@" RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 4 --no-cpu-baseline --force-allreduce 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('samples/s', d['value'], 'ms/step', d['ms_per_step'], 'host enqueue ms', d.get('host_enqueue_ms_per_step'))
" >> $OUT 2>&1; }
run SSM_WINO_SKIP=conv6.0,conv6.1
run A=1
run SSM_WINO_SKIP=conv6.0,conv6.1
run A=1
cat $OUT
