#!/bin/bash
# same-box A/B of the training step: split-K launches on the tile configuration with the fewest workgroups (default) vs the cost model's pick (r5)
OUT=${1:-gpurun_out/r35_split_tiles_ab.txt}
run() { echo "== $*" >> $OUT; env "$@" RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 4 --no-cpu-baseline --force-allreduce 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('samples/s', d['value'], 'ms/step', d['ms_per_step'], 'host enqueue ms', d.get('host_enqueue_ms_per_step'))
" >> $OUT 2>&1; }
run SSM_WINO_SPLIT_TILES=cost
run SSM_WINO_SPLIT_TILES=fewest
run SSM_WINO_SPLIT_TILES=cost
run SSM_WINO_SPLIT_TILES=fewest
cat $OUT
