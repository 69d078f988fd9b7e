#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (tools/pmc_conv.sh / pmc_kernel.sh: one counter_collection.csv per pass) into
profiles/<tag>_pmc_<label>.json: per kernel name the per-dispatch mean of every counter and of the dispatch duration, plus
MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x clock) at the nominal 2.4 GHz and at the clock
GRBM_GUI_ACTIVE implies where collected (MI355X_MICROARCH.md, DVFS give-back: effective clock = GRBM_GUI_ACTIVE / 8 / wall).
usage: python tools/summarize_pmc.py <pmc_dir> <kernel regex> <out json>"""
import collections
import csv
import glob
import json
import re
import sys

src, rx, out = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(f"{src}/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not rx.search(name):
            continue
        key = name.split("(")[0].replace("void ", "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        did = (f, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did)
            dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {}
for key, ctrs in acc.items():
    d = sorted(dur[key])
    ns = d[len(d) // 2]                                   # median dispatch duration under the profiler
    row = {"dispatches": len(d), "median_duration_us": round(ns / 1e3, 1)}
    for c, vals in sorted(ctrs.items()):
        row[c] = round(sum(vals) / len(vals), 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row:
        simd = 4 * 256
        row["mfma_pipe_util_at_2.4GHz"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / (simd * ns * 2.4), 4)
        if row.get("GRBM_GUI_ACTIVE"):
            ghz = row["GRBM_GUI_ACTIVE"] / 8.0 / ns            # summed over the 8 XCDs
            row["effective_clock_GHz"] = round(ghz, 3)
            row["mfma_pipe_util_at_effective_clock"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / (simd * ns * ghz), 4)
    if "SQ_LDS_BANK_CONFLICT" in row and row.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_conflict_fraction"] = round(row["SQ_LDS_BANK_CONFLICT"] / row["SQ_LDS_IDX_ACTIVE"], 4)
    res[key] = row
json.dump({"source": f"rocprofv3 --pmc passes in {src}, kernels matching /{sys.argv[2]}/", "kernels": res}, open(out, "w"), indent=1)
for k, v in res.items():
    print(k, {a: b for a, b in v.items() if a in ("median_duration_us", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "mfma_pipe_util_at_2.4GHz", "lds_conflict_fraction")})
