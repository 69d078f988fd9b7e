cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python - <<'PY'
import json,subprocess,sys
out=subprocess.run([sys.executable,"bench.py","--steps","30","--warmup","3","--no-cpu-baseline","--no-units"],capture_output=True,text=True).stdout
d=json.loads(out.strip().splitlines()[-1])
print(d["ms_per_step"])
for k,v in d["kernel_ms_per_step"].items(): print(k,v)
PY
