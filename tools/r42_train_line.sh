#!/bin/bash
# the training line of the default bench (config 3, f32w, exchange forced), n times: tools/r42_train_line.sh <out> [n] [ENV=...]
OUT=${1:-gpurun_out/r42_train_line.txt}; N=${2:-3}; shift 2
for i in $(seq $N); do
env "$@" RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 4 --no-cpu-baseline --force-allreduce 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$*', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'host enqueue ms', d.get('host_enqueue_ms_per_step'))
" >> $OUT 2>&1
done
