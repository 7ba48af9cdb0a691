#!/bin/bash
# Matrix-pipe busy counters of the chain's kernels (one PMC pass): is gcorr_kernel short of matrix work or short of pipe?
# usage (repo root, GPU box): bash tools/pmc_mfma.sh [split_profile.py args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/mfma
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p -- python3 $R/tools/split_profile.py ${1:-module0} > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
(cd $R && python3 tools/pmc_sq.py $O/p gtables gcorr pixel_adc)
rm -rf $O/p
