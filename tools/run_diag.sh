cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ddp_gpu.py -x -q -m gpu > gpurun_out/ddpblk.log 2>&1; tail -25 gpurun_out/ddpblk.log | cut -c1-250
