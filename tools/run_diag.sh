cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_diag31.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/gputest_diag31.log
for i in 1 2; do python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c100-200; done
