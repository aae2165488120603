set -x
mkdir -p gpurun_out
timeout 3400 python -m pytest tests/ -q -m gpu > gpurun_out/r2q_full_tests.log 2>&1; tail -n 6 gpurun_out/r2q_full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2q_smoke.log 2>&1; tail -n 4 gpurun_out/r2q_smoke.log
