export SEGHIERO_BENCH_BACKEND=gloo SEGHIERO_BENCH_ONE_DEVICE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2
RANK=1 LOCAL_RANK=1 SEGHIERO_TRACE=1 timeout -k 10 300 python -X faulthandler bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 $BENCH_EXTRA > gpurun_out/r1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 $BENCH_EXTRA > gpurun_out/r0.log 2>&1
echo "rank0 exit $?"
wait $P1; echo "rank1 exit $?"
echo ---- r1; grep -v "amdgpu\|^\[W" gpurun_out/r1.log | tail -15 | cut -c1-300
echo ---- r0; grep -v "amdgpu\|^\[W" gpurun_out/r0.log | tail -5 | cut -c1-400
