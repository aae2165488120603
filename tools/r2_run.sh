set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_conv_f32.py tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_backward.py -q -m gpu > gpurun_out/r3d_tests.log 2>&1; tail -n 3 gpurun_out/r3d_tests.log
timeout 600 python tools/tune_conv_f32.py 14 2 > gpurun_out/r3d_tune_b14.log 2>&1; tail -n 1 gpurun_out/r3d_tune_b14.log
timeout 600 python tools/tune_conv_f32.py 2 1 > gpurun_out/r3d_tune_b2.log 2>&1; tail -n 1 gpurun_out/r3d_tune_b2.log
timeout 2400 python bench.py --detail gpurun_out/r3d_detail.json > gpurun_out/r3d_bench_line.json 2> gpurun_out/r3d_bench_err.log
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r3d_bench_line.json').read())
print(l['value'], l['dtype'], l['roofline']['achieved'], l['roofline']['frac'], l['roofline'].get('shader_clock',{}).get('ghz'), l['roofline']['detail']['achieved_in_kernel'], {k:v['value'] for k,v in l.get('modes',{}).items()}, l['parity']['max_abs_vs_oracle'])
PY
