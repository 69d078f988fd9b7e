# usage (GPU box): bash tools/wg_target_sweep.sh "0 32 64"   -- wgrad: max tiles per slice for the one-slice-per-XCD mapping
for v in $1; do
  echo "== SEGHIERO_WG_XCD_TILES=$v"
  SEGHIERO_WG_XCD_TILES=$v timeout -k 10 200 python tools/bench_conv.py 2>&1 | grep -v "amdgpu" | sed 's/fprop.*| wgrad/wgrad/'
done
