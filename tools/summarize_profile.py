#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh run into profiles/<tag>_*.{csv,json}.

HBM traffic per kernel family from the PMC passes: FETCH_SIZE / WRITE_SIZE are reported in KiB (x1024 -> bytes) and, on
gfx950, FETCH_SIZE counts a wide coalesced streaming read at exactly half its bytes (MI355X_MICROARCH.md, HBM section),
so the read side is doubled before comparing with algorithmic bytes; WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01b"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"          # (on the GPU box: gpurun_out/profiles_out, merged back and copied to profiles/)
src = f"gpurun_out/prof_{tag}"
stats = glob.glob(f"{src}/trace/runc/*_kernel_stats.csv")[0]
shutil.copy(stats, f"{dst}/{tag}_kernel_stats.csv")


def family(name):
    for key, fam in (("conv_x6_kernel<0", "conv_fprop_x6"), ("conv_x6_kernel<1", "conv_dgrad_x6"), ("conv_wgrad_x6", "conv_wgrad_x6"),
                     ("conv_x6p_kernel<0", "conv_fprop_x6"), ("conv_x6p_kernel<1", "conv_dgrad_x6"),
                     ("conv_gemm_kernel<0", "conv_fprop_f32"), ("conv_gemm_kernel<1", "conv_dgrad_f32"),
                     ("conv_gemm_kernel<2", "conv_wgrad_f32")):
        if key in name:
            return fam
    return name.split("(")[0].replace("void ", "")[:48]


PMC_STEPS = 7.0
out = collections.defaultdict(lambda: dict(launches=0, fetch_kib=0.0, write_kib=0.0))
for which, col in (("fetch", "fetch_kib"), ("write", "write_kib")):
    f = glob.glob(f"{src}/{which}/runc/*_counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    # bench ran warmup 1 + steps 1 + host-timing 3 + instrumented 2 = 7 identical steps: average per step
    for r in rows:
        fam = family(r["Kernel_Name"])
        out[fam][col] += float(r["Counter_Value"]) / PMC_STEPS
        if which == "fetch":
            out[fam]["launches"] += 1.0 / PMC_STEPS
res = {}
for fam, d in out.items():
    rd, wr = 2.0 * d["fetch_kib"] * 1024, d["write_kib"] * 1024
    res[fam] = dict(launches_per_step=round(d["launches"], 1), hbm_read_bytes_per_step=rd, hbm_write_bytes_per_step=wr,
                    hbm_bytes_per_launch=(rd + wr) / max(d["launches"], 1e-9))
# achieved HBM rate per family: PMC bytes per step / kernel time per step from the trace pass (its stats file covers TRACE_STEPS steps:
# 2 warm-up + 4 timed + 3 host-timing + 2 instrumented); 8 TB/s is the guide's HBM peak, ~6.3 TB/s what a plain streaming kernel reaches
TRACE_STEPS = 11.0
ms = collections.defaultdict(float)
for r in csv.DictReader(open(stats)):
    ms[family(r["Name"])] += float(r["TotalDurationNs"]) / TRACE_STEPS / 1e6
for fam, d in res.items():
    t = ms.get(fam, 0.0)
    d["ms_per_step"] = round(t, 4)
    d["achieved_TBps"] = round((d["hbm_read_bytes_per_step"] + d["hbm_write_bytes_per_step"]) / (t * 1e-3) / 1e12, 3) if t > 0 else None
top = dict(sorted(res.items(), key=lambda kv: -(kv[1]["hbm_read_bytes_per_step"] + kv[1]["hbm_write_bytes_per_step"]))[:25])
with open(f"{dst}/{tag}_hbm_rates.txt", "w") as fh:
    fh.write("# HBM bytes per step (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE) / kernel time per step (rocprofv3 --kernel-trace --stats), by kernel family\n")
    fh.write("# peak 8.0 TB/s (MI355X_MICROARCH.md); the MFMA-bound conv families are listed for their traffic, not as a bandwidth claim\n")
    fh.write(f"{'family':34s} {'launches':>8s} {'ms/step':>8s} {'GB/step':>8s} {'TB/s':>6s} {'of 8 TB/s':>9s}\n")
    for fam, d in top.items():
        gb = (d["hbm_read_bytes_per_step"] + d["hbm_write_bytes_per_step"]) / 1e9
        r = d["achieved_TBps"]
        fh.write(f"{fam[:34]:34s} {d['launches_per_step']:8.1f} {d['ms_per_step']:8.3f} {gb:8.2f} {(r if r is not None else 0):6.2f} {(r / 8.0 if r else 0):9.2f}\n")
import hashlib
import os
_h = hashlib.sha1()
for _f in sorted(glob.glob("seghiero_amd/csrc/*.hip") + glob.glob("seghiero_amd/csrc/*.h")):
    _h.update(open(_f, "rb").read())
json.dump({"csrc_sha": _h.hexdigest()[:12], "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `python bench.py --steps 1 --warmup 1` ({tag}); FETCH_SIZE doubled "
                     "(gfx950 wide-read correction), KiB -> bytes", "per_kernel_family": top}, open(f"{dst}/{tag}_hbm_traffic.json", "w"), indent=1)
for fam, d in list(top.items())[:14]:
    print(f"{fam:28s} launches/step {d['launches_per_step']:6.1f}  read {d['hbm_read_bytes_per_step']/1e9:7.2f} GB  write {d['hbm_write_bytes_per_step']/1e9:7.2f} GB")

# the north-star unit's own trace (tools/profile_round.sh step 4), if it was taken
import os
au = glob.glob(f"{src}/aspp/*/*_kernel_stats.csv")
if au:
    shutil.copy(au[0], f"{dst}/{tag}_aspp_unit_kernel_stats.csv")
    if os.path.exists(f"gpurun_out/aspp_unit_{tag}.json"):
        txt = open(f"gpurun_out/aspp_unit_{tag}.json").read()
        open(f"{dst}/{tag}_aspp_unit.json", "w").write(txt[txt.index("{"):])
