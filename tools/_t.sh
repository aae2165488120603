set -e
timeout -k 10 1000 python -m pytest tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do
SSM_FUSED_LOSS=$v timeout -k 10 300 python bench.py --no-configs --mode train --precision f32w --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/_e.txt | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('SSM_FUSED_LOSS=$v', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
