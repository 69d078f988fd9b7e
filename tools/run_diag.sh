cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_loss_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "three_level or rmi or config4" 2>&1 | tail -4
python tools/time_config.py C4 2>&1 | grep -v amdgpu | head -16
