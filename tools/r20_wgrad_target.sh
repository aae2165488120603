#!/bin/bash
# in-step sweep of the DIRECT weight-gradient kernel's workgroups per launch now that the big 3x3 layers run in the Winograd domain
OUT=${1:-gpurun_out/r20_wgrad_target.txt}
for t in 512 256 384 768 512; do
  echo "== SSM_WGRAD_TARGET=$t" >> $OUT
  SSM_WGRAD_TARGET=$t RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 4 --no-cpu-baseline --force-allreduce 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith(chr(123)):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])
" >> $OUT 2>&1
done
cat $OUT
