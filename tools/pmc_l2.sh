#!/bin/bash
# L2 hit rate and fabric-side read bytes of the chain's kernels on one response table (separate PMC passes)
# usage (repo root, GPU box): bash tools/pmc_l2.sh <config> <response>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; RESP=$2
O=$R/gpurun_out/l2_${CFG}_$RESP
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --config $CFG --response $RESP --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit -- python3 $B > $O/hit.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $B > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/tcp -- python3 $B > $O/tcp.log 2>&1 || echo "tcp pass failed"
python3 - <<PY
import csv, glob, collections
for sub in ("hit", "fetch", "tcp"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
        print(sub, k, {c: "%.4g per dispatch" % (v / max(n[(k, c)], 1)) for c, v in d.items()})
PY
rm -rf $O/hit $O/fetch $O/tcp
