set -x
python tools/bench_layers.py 7 > gpurun_out/r2a_layers_b7.log 2>&1
python tools/bench_layers.py 1 > gpurun_out/r2a_layers_b1.log 2>&1
python tools/clock_under_load.py --f32 conv1b conv2b conv5b conv8a conv8b conv10a conv11a fuse_conv > gpurun_out/r2a_clock_f32.log 2>&1
python bench.py --precision f32 --steps 10 --warmup 2 --no-cpu-baseline --detail gpurun_out/r2a_f32_detail.json > gpurun_out/r2a_f32_line.json 2> gpurun_out/r2a_f32_err.log
bash tools/pmc_conv_f32.sh gpurun_out/r2a_pmc_f32 7 > gpurun_out/r2a_pmc_f32.log 2>&1
tail -3 gpurun_out/r2a_layers_b7.log; cat gpurun_out/r2a_clock_f32.log; cat gpurun_out/r2a_f32_line.json | cut -c1-600
