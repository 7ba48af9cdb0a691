#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE: separate runs) of the bench for one configuration -> profiles/r04_traffic.json
# usage (from the repo root, on the GPU box): bash tools/pmc_passes.sh <config> [extra bench flags]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
O=$R/gpurun_out/pmc_$CFG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B > $O/pmc_write.log 2>&1 || exit 1
(cd $R && python3 tools/pmc_traffic.py $CFG $O/pmc_fetch $O/pmc_write > $O/traffic.log 2>&1 && cp profiles/r04_traffic.json $O/r04_traffic.json) || exit 1
rm -rf $O/pmc_fetch $O/pmc_write
tail -8 $O/traffic.log
