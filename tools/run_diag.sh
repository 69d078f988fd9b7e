cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do
for r in 1 1000 0.5 2; do
SEGHIERO_DEFER_RATIO=$r python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c1-200 | sed "s/^/ratio $r: /" | cut -c1-20,100-200
done
done
