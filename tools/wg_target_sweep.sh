# usage (GPU box): bash tools/wg_target_sweep.sh "512 640 768 1024"   -- wgrad split-K block-count target experiment
for v in $1; do
  echo "== target $v"
  SEGHIERO_WG_TARGET=$v timeout -k 10 200 python tools/bench_conv.py 2>&1 | grep -v "amdgpu" | sed 's/fprop.*| wgrad/wgrad/'
done
