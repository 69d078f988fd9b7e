cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "dwconv or head or train_steps or config2" > gpurun_out/gputest_diag25.log 2>&1; echo "tests exit $?"; tail -8 gpurun_out/gputest_diag25.log
for i in 1 2; do
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c100-200 | sed "s/^/new /"
done
