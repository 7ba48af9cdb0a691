"""Sum per-kernel PMC counters of a rocprofv3 --pmc run (csv output): python tools/pmc_sq.py <dir> [kernel substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:] or ["gcorr", "gtables", "mac_", "qweights", "pixel_adc"]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if not any(s in k for s in subs):
                continue
            name = k.split("(")[0]
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[name].add(row["Dispatch_Id"])
for k, v in acc.items():
    print(k, "dispatches", len(cnt[k]))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:.4g}" + (f"   ({x / wc:.3f} of wave cycles)" if wc and c != "SQ_WAVE_CYCLES" else ""))
