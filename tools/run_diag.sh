cd "$GRAFT_REPO_ROOT"; python tests/diag/fuse_ab.py 2>&1 | grep -v "amdgpu\|Warning\|warn"
