#!/usr/bin/env python3
"""Reduces a rocprofv3 --pmc counter_collection CSV to the igemm dispatches only (config, fast flag, counter value,
duration in ns, in dispatch order) so that the multi-MB CSV does not have to be kept.
    python3 tools/traffic_extract.py <dir with p_counter_collection.csv>  ->  <dir>/igemm_dispatches.json"""
import csv, json, re, sys
d = sys.argv[1]
rows = []
with open(f"{d}/p_counter_collection.csv") as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        m = re.search(r"igemm_kernel<[^>]*Cfg<([^>]*)>, (true|false)", n)
        if m:
            rows.append([m.group(1).replace(" ", ""), m.group(2), float(r["Counter_Value"]),
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
        elif "igemm8_kernel" in n:
            rows.append(["4,2,4,8:8phase", "true", float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
        elif "ffn320_kernel" in n:                           # the fused GEGLU feed-forward: counted with the family (ops.ffn_geglu records a shape too)
            rows.append(["ffn320:fused", "true", float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
        elif "lnlin320_kernel" in n:                         # LayerNorm + Q | K | V projection in one launch (ops.ln_linear records a shape too)
            rows.append(["lnlin320:fused", "true", float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
        elif "igemm10_kernel" in n:
            rows.append(["4,2,4,10:10phase", "true", float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
json.dump(rows, open(f"{d}/igemm_dispatches.json", "w"))
print(len(rows), "igemm dispatches")
