set -e
timeout -k 10 600 python -m pytest $(grep -ln "PackBatch32\|pack32" tests/*.py) -x -q -m gpu 2>&1 | tail -2
timeout -k 10 1000 python -m pytest tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do
SSM_PACK_TILES=$v timeout -k 10 300 python bench.py --no-configs --mode train --precision f32w --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/_e.txt | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('SSM_PACK_TILES=$v', d['value'], d['ms_per_step'])"
done
