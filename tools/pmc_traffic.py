"""HBM bytes per launch of the chain's kernels from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  Both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read (same guide, HBM), so it is doubled.  Writes profiles/r04_traffic.json (LDSIM_TRAFFIC_FILE names another), which bench.py reads for
`roofline.traffic`, and copies the per-kernel averages next to it.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ... (same command)
  python tools/pmc_traffic.py module0 gpurun_out/pmc_fetch gpurun_out/pmc_write
(without --output-format csv rocprofv3 leaves a *_results.db instead; both are read)

  python tools/pmc_traffic.py --kernel-stats gpurun_out/r02_stats/ks_results.db profiles/r02_kernel_stats_module0.csv
turns the sqlite output of `rocprofv3 --kernel-trace --stats` into the kernel_stats.csv table (same columns, ns).
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
    # rocprofv3 without --output-format csv writes one sqlite file (rocpd): the same rows sit in its counters_collection view
    for f in glob.glob(os.path.join(directory, "**", "*_results.db"), recursive=True):
        import sqlite3
        with sqlite3.connect(f) as db:
            for kname, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
                name = re.sub(r"^void ", "", kname).split("(")[0]
                tot[name] = tot.get(name, 0.0) + float(value)
                cnt[name] = cnt.get(name, 0) + 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def kernel_stats(db_path, out_csv):
    import sqlite3
    import statistics
    per = {}
    with sqlite3.connect(db_path) as db:
        for name, dur in db.execute("select name, duration from kernels"):
            per.setdefault(name, []).append(float(dur))
    total = sum(sum(v) for v in per.values())
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), int(sum(v)), round(sum(v) / len(v), 6), round(100.0 * sum(v) / total, 2), int(min(v)), int(max(v)),
                        round(statistics.stdev(v), 6) if len(v) > 1 else 0.0])
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print(f"{name.split('(')[0][:50]:50s} {len(v):4d} launches  {sum(v) / len(v) / 1e6:9.3f} ms avg")


def main():
    if sys.argv[1] == "--kernel-stats":
        return kernel_stats(sys.argv[2], sys.argv[3])
    config, d_fetch, d_write = sys.argv[1:4]
    fetch, write = per_kernel(d_fetch, "FETCH_SIZE"), per_kernel(d_write, "WRITE_SIZE")
    path = os.path.join(REPO, "profiles", os.environ.get("LDSIM_TRAFFIC_FILE", "r04_traffic.json"))
    tab = json.load(open(path)) if os.path.exists(path) else {}
    entry = {}
    # a kernel launched several times per chain launch (gcorr_kernel: once per LDS class) is summed over them: "per launch" means
    # per chain launch
    # (make_keys_kernel runs once per chain launch; the FEE stage is two list launches since round 4)
    chain_f, chain_w = fetch.get("make_keys_kernel", (0.0, 0))[1], write.get("make_keys_kernel", (0.0, 0))[1]
    for k in sorted(set(fetch) | set(write)):
        f_kib, nf = fetch.get(k, (0.0, 0))
        w_kib, nw = write.get(k, (0.0, 0))
        per_f = nf / chain_f if chain_f and nf > chain_f else 1.0
        per_w = nw / chain_w if chain_w and nw > chain_w else 1.0
        entry[k] = {"bytes_per_launch": 2.0 * f_kib * per_f * 1024 + w_kib * per_w * 1024, "fetch_size_kib_avg": f_kib,
                    "write_size_kib_avg": w_kib, "dispatches_per_chain_launch": per_f,
                    "launches_profiled": [nf, nw],
                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x 2 per the gfx950 correction"}
    tab[os.environ.get("LDSIM_TRAFFIC_KEY", config)] = entry
    with open(path, "w") as fh:
        json.dump(tab, fh, indent=1, sort_keys=True)
    for k, e in sorted(entry.items(), key=lambda kv: -kv[1]["bytes_per_launch"])[:8]:
        print(f"{config:8s} {k:40s} {e['bytes_per_launch'] / 1e9:9.3f} GB per launch")


if __name__ == "__main__":
    main()
