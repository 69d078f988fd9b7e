cd "$GRAFT_REPO_ROOT"; python tests/diag/aspp_host.py 2>&1 | grep -v amdgpu | head -50
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x --timeout=600 -k "head" 2>&1 | tail -8
