set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_model.py -q -m gpu -k "two_pairs_per_pass or interpolate_many" > gpurun_out/r2r_tests.log 2>&1; tail -n 3 gpurun_out/r2r_tests.log
timeout 2400 python bench.py --detail gpurun_out/r2r_detail.json > gpurun_out/r2r_bench_line.json 2> gpurun_out/r2r_bench_err.log
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r2r_bench_line.json').read())
print(l['value'], l['dtype'], l['roofline']['achieved'], l['roofline']['frac'], l['roofline'].get('shader_clock'), {k:v['value'] for k,v in l.get('modes',{}).items()}, l['parity']['max_abs_vs_oracle'], l['cpu_baseline']['value'])
PY
tail -n 3 gpurun_out/r2r_bench_err.log
