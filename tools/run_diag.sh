cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python tools/time_config.py C4 2>&1 | grep -v amdgpu
python tools/time_config.py C5 2>&1 | grep -v amdgpu
