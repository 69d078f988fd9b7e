#!/usr/bin/env python3
"""The north-star unit on its own: python tools/bench_aspp_branch.py  (prints seghiero_amd.units.measure() for the fp32-accurate path
and for bf16 compute mode)."""
import json
import sys
sys.path.insert(0, ".")
from seghiero_amd import units
print(json.dumps({"aspp_ds_branch": units.measure(), "aspp_ds_branch_bf16": units.measure(bf16=True)}, indent=1))
