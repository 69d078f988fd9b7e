# usage (GPU box): bash tools/x6_variant_sweep.sh "0 6" "sep1 sep0 l2.conv3"   -- tile-variant experiments (SEGHIERO_X6_VARIANT)
for v in $1; do
  echo "== variant $v"
  for sh in $2; do
    SEGHIERO_X6_VARIANT=$v timeout -k 10 120 python tools/bench_conv.py "$sh" 2>&1 | grep -v "amdgpu\|TOTAL"
  done
done
