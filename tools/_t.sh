export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611
for i in 1 2; do
timeout -k 10 300 python bench.py --gpus 1 --no-configs --mode train --precision f32w --steps 20 --warmup 3 --force-allreduce --no-cpu-baseline > gpurun_out/_o.txt 2> gpurun_out/_e.txt; echo "rc $?"; tail -c 1500 gpurun_out/_o.txt | cut -c1-1500; tail -5 gpurun_out/_e.txt
done
