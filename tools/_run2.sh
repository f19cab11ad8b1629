export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 600 python bench.py --force-transport --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/ft.out 2> gpurun_out/ft.err; echo "rc=$?"
cut -c1-300 gpurun_out/ft.out | tail -5; echo ---; tail -15 gpurun_out/ft.err | cut -c1-200
