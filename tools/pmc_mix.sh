#!/bin/bash
# Instruction mix of the chain's kernels (PMC passes of tools/split_profile.py): which issue port is the busiest?
# usage (repo root, GPU box): bash tools/pmc_mix.sh [config]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/mix
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python3 $R/tools/split_profile.py ${1:-module0} > $O/run1.log 2>&1 || { tail -5 $O/run1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/p2 -- python3 $R/tools/split_profile.py ${1:-module0} > $O/run2.log 2>&1 || { tail -5 $O/run2.log; exit 1; }
(cd $R && python3 tools/pmc_sq.py $O/p1 gtables_wave gcorr pair_setup; python3 tools/pmc_sq.py $O/p2 gtables_wave gcorr pair_setup)
rm -rf $O/p1 $O/p2
