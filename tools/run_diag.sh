cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout=600 2>&1 | tail -4
SEGHIERO_BENCH_FUSED=1 timeout -k 10 400 python tools/bench_conv.py > gpurun_out/convbench_fused2.txt 2>&1
cat gpurun_out/convbench_fused2.txt | grep -v amdgpu.ids | cut -c1-250
