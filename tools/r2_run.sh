set -x
mkdir -p gpurun_out
for sg in 0 1; do
SSM_CONV_STAGGER=$sg timeout 600 python tools/bench_layers.py 7 > gpurun_out/r2g_layers_b7_stag$sg.log 2>&1; tail -n 1 gpurun_out/r2g_layers_b7_stag$sg.log
SSM_CONV_STAGGER=$sg timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-io --modes '' --streams 3 --pairs-per-batch 2 --detail gpurun_out/r2g_detail_stag$sg.json > gpurun_out/r2g_f32_stag$sg.json 2>> gpurun_out/r2g_err.log
cut -c1-200 gpurun_out/r2g_f32_stag$sg.json
done
timeout 1500 python -m pytest tests/test_hip_conv_f32.py tests/test_hip_model.py -x -q -m gpu > gpurun_out/r2g_tests.log 2>&1; tail -n 5 gpurun_out/r2g_tests.log
