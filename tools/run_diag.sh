cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c100-200 | sed 's/^/base /'
SEGHIERO_BNB_RESIDUAL=1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-units 2>/dev/null | cut -c100-200 | sed 's/^/resid /'
done
SEGHIERO_BNB_RESIDUAL=1 python - <<'PY'
import json,subprocess,sys
out=subprocess.run([sys.executable,"bench.py","--steps","20","--warmup","3","--no-cpu-baseline","--no-units"],capture_output=True,text=True).stdout
d=json.loads(out.strip().splitlines()[-1])
for k,v in d["kernel_ms_per_step"].items(): print(k,v)
PY
