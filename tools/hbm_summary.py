"""Mean FETCH_SIZE / WRITE_SIZE per launch of every kernel with at least 20 launches in the rocprofv3 --pmc passes
tools/gpu_round5.sh hbm wrote (directories <R>/hbm_<workload>_<counter>/).  Raw counter units (KiB); the gfx950
correction of the guide is applied by whoever reads this (tools/hbm_traffic.py).  Usage: python3 tools/hbm_summary.py <R>"""
import collections
import csv
import glob
import json
import os
import sys

R = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(R, "hbm_*_*_SIZE"))):
    base = os.path.basename(d)
    w, c = base[4:].rsplit("_", 2)[0], "_".join(base.rsplit("_", 2)[1:])
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if len(v) >= 20:
                v = v[len(v)//5:]          # steady state: drop the warm-up launches
                out.setdefault(w, {}).setdefault(k, {})[c] = {"launches": len(v), "mean_KiB": sum(v)/len(v), "min_KiB": min(v), "max_KiB": max(v)}
json.dump(out, sys.stdout, indent=1)
print()
