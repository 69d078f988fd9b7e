cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_loss_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/gputest_diag17.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/gputest_diag17.log
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/bench_diag17.json 2> gpurun_out/bench_diag17.err; echo "bench exit $?"; cut -c1-300 gpurun_out/bench_diag17.json
