cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
bash tests/diag/run2ranks.sh 2>&1 | tail -12
BENCH_EXTRA=--syncbn bash tests/diag/run2ranks.sh 2>&1 | tail -6
