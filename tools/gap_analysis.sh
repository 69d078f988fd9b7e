#!/bin/bash
# Idle time between consecutive kernels of a step (single stream):  bash tools/gap_analysis.sh <tag> <cfg> <batch> [mode]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gap_$TAG; rm -rf $OUT; mkdir -p $OUT
SEGHIERO_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python tools/time_config.py "$@" > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
n = len(rows)
rows = rows[n // 3: 2 * n // 3]           # the middle third of the run: steady-state steps
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({busy / span:.3f}), idle between kernels {sum(pos) / 1e6:.2f} ms "
      f"= {sum(pos) / len(rows) / 1e3:.2f} us per launch; gaps > 20 us: {sum(1 for g in pos if g > 20000)} totalling {sum(g for g in pos if g > 20000) / 1e6:.2f} ms")
import collections
after = collections.defaultdict(lambda: [0, 0])
for i, g in enumerate(gaps):
    if g > 0:
        k = rows[i + 1][2].split("(")[0][:50]
        after[k][0] += g; after[k][1] += 1
for k, (t, c) in sorted(after.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"   idle before {k:52s} {t / 1e6:7.3f} ms over {c:5d} launches = {t / c / 1e3:6.2f} us each")
PY
rm -rf $OUT/*/
