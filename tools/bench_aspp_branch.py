#!/usr/bin/env python3
"""The north-star unit on its own: python tools/bench_aspp_branch.py  (prints seghiero_amd.units.measure())."""
import json
import sys
sys.path.insert(0, ".")
from seghiero_amd import units
print(json.dumps(units.measure(), indent=1))
