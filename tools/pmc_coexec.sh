#!/bin/bash
# Do vector and f64 matrix instructions execute together in gcorr_kernel?  (SQ_VALU_MFMA_COEXEC_CYCLES against the two busy counters)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/coexec
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/p -- python3 $R/tools/split_profile.py ${1:-module0} > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
(cd $R && python3 tools/pmc_sq.py $O/p gcorr)
rm -rf $O/p
