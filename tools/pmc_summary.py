#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel name.   python3 tools/pmc_summary.py <dir with *counter_collection.csv>
(% of SIMD cycles = value / (GRBM_GUI_ACTIVE x 128): GRBM_GUI_ACTIVE is summed over the 8 XCDs, 4 SIMDs x 256 CUs / 8)"""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for path in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
for k, c in agg.items():
    print(f"{k}  launches {n[k]}")
    g = c.get("GRBM_GUI_ACTIVE", 0.0)
    for name, v in sorted(c.items()):
        pct = f"   = {100 * v / (g * 128):5.1f} % of SIMD cycles" if g and name != "GRBM_GUI_ACTIVE" else ""
        print(f"   {name:28s} {v:16.0f}{pct}")
