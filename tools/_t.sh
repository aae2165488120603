for v in 0 1 0 1; do
  SSM_WINO_SPLITK=$v timeout -k 10 200 python bench.py --no-configs --no-cpu-baseline --modes "" --no-io --steps 10 2>>gpurun_out/_e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SSM_WINO_SPLITK=$v', d['value'], d['config']['ms_per_pair'], {k: v['ms_per_pair_in_kernel'] for k,v in d['roofline']['families'].items()})"
done
