cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/gputest_diag36.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/gputest_diag36.log
for i in 1 2 3; do
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c100-200 | sed 's/^/new /'
(cd .ab_old && python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c100-200 | sed 's/^/old /')
done
python tools/shape_table.py > gpurun_out/shape_table_new.txt 2>&1
(cd .ab_old && python tools/shape_table.py > ../gpurun_out/shape_table_old.txt 2>&1)
