cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in sep1.pw sep0.pw l2.conv2 l3.conv3 proj; do
  python tools/bench_conv.py $s 2>&1 | grep -v "amdgpu\|TOTAL"
  SEGHIERO_LIB=$PWD/seghiero_amd/csrc/alt/libseghiero_ablate.so python tools/bench_conv.py $s 2>&1 | grep -v "amdgpu\|TOTAL" | sed 's/^/   ablateB: /'
done
