"""HBM bytes per launch of the chain's kernels from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  Both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read (same guide, HBM), so it is doubled.  Writes profiles/r02_traffic.json, which bench.py reads for
`roofline.traffic`, and copies the per-kernel averages next to it.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ... (same command)
  python tools/pmc_traffic.py module0 gpurun_out/pmc_fetch gpurun_out/pmc_write
"""
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(directory, counter):
    tot, cnt = {}, {}
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0]
                tot[name] = tot.get(name, 0.0) + float(row["Counter_Value"])
                cnt[name] = cnt.get(name, 0) + 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    config, d_fetch, d_write = sys.argv[1:4]
    fetch, write = per_kernel(d_fetch, "FETCH_SIZE"), per_kernel(d_write, "WRITE_SIZE")
    path = os.path.join(REPO, "profiles", "r02_traffic.json")
    tab = json.load(open(path)) if os.path.exists(path) else {}
    entry = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, nf = fetch.get(k, (0.0, 0))
        w_kib, nw = write.get(k, (0.0, 0))
        entry[k] = {"bytes_per_launch": 2.0 * f_kib * 1024 + w_kib * 1024, "fetch_size_kib_avg": f_kib, "write_size_kib_avg": w_kib,
                    "launches_profiled": [nf, nw],
                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x 2 per the gfx950 correction"}
    tab[config] = entry
    with open(path, "w") as fh:
        json.dump(tab, fh, indent=1, sort_keys=True)
    for k, e in sorted(entry.items(), key=lambda kv: -kv[1]["bytes_per_launch"])[:8]:
        print(f"{config:8s} {k:40s} {e['bytes_per_launch'] / 1e9:9.3f} GB per launch")


if __name__ == "__main__":
    main()
