export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611
for i in 1 2; do
timeout -k 10 300 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 3 --force-allreduce --no-cpu-baseline > gpurun_out/_o.txt 2> gpurun_out/_e.txt; python -c "
import json; d=json.loads([l for l in open('gpurun_out/_o.txt').read().splitlines() if l.startswith('{')][-1]); print('forced RCCL buckets, one-rank group:', d['value'], d['ms_per_step'], {k:v for k,v in d['allreduce'].items() if k!='note'}, 'host enqueue', d['host_enqueue_ms_per_step'], d['time_split_ms_per_step'])"
done
