#!/usr/bin/env python3
"""Median per-launch value of every counter of every kernel in the passes TAG_p* under DIR (tools/sq_passes.sh)."""
import csv, glob, json, os, statistics, sys
d, tag = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(os.path.join(d, tag + "_p*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].replace("void ", "").split("(")[0]
            out.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
res = {k: {c: statistics.median(v) for c, v in cs.items()} for k, cs in out.items()}
for k, cs in res.items():
    cs["_launches"] = max(len(v) for v in out[k].values())
print(json.dumps(res, indent=1))
