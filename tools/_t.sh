set -e
timeout -k 10 600 python -m pytest tests/test_hip_conv_f32.py tests/test_hip_wino.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 1000 python -m pytest tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --no-configs --mode train --precision f32w --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/_e.txt | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('direct-form split-K:', d['value'], d['ms_per_step'], d['time_split_ms_per_step'])"
done
