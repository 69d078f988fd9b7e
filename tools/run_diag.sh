cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in sep1.pw l2.conv2 l3.conv3 proj l3.conv2; do
  python tools/bench_conv.py "$s" 2>&1 | grep -v "amdgpu\|TOTAL"
  SEGHIERO_LIB=$PWD/seghiero_amd/csrc/alt/libseghiero_m16.so python tools/bench_conv.py "$s" 2>&1 | grep -v "amdgpu\|TOTAL" | sed 's/^/   m16 : /'
done
for i in 1 2; do
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c150-185 | sed 's/^/base /'
SEGHIERO_LIB=$PWD/seghiero_amd/csrc/alt/libseghiero_m16.so python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c150-185 | sed 's/^/m16  /'
done
