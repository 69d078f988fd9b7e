#!/bin/bash
# Per-kernel time of one configuration's step, filtered:  bash tools/prof_kernels.sh <tag> <regex> <time_config args...>
TAG=$1; RE=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pk_$TAG; rm -rf $OUT; mkdir -p $OUT
SEGHIERO_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/time_config.py "$@" > $OUT/run.log 2>&1
python - "$OUT" "$RE" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:80]:82s} {int(r['Calls']):5d} calls  {float(r['AverageNs']) / 1e3:9.1f} us avg  {float(r['TotalDurationNs']) / 34e6:7.3f} ms/step")
PY
rm -rf $OUT/*/
