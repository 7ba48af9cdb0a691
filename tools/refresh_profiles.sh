#!/bin/bash
# One measurement pass on the GPU box: PMC traffic, kernel stats, SQ counters, the bench lines.  Everything lands in gpurun_out/rp/.
# usage (from the repo root): bash tools/refresh_profiles.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/rp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
echo "pmc fetch" > $O/progress; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B > $O/pmc_fetch.log 2>&1 || exit 1
echo "pmc write" >> $O/progress; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B > $O/pmc_write.log 2>&1 || exit 1
(cd $R && python3 tools/pmc_traffic.py module0 $O/pmc_fetch $O/pmc_write > $O/traffic.log 2>&1 && cp profiles/r04_traffic.json $O/r04_traffic.json) || exit 1
rm -rf $O/pmc_fetch $O/pmc_write
# the same two passes on the full-support ("dense") table (matrix form too since round 4: gform_max_support unlimited)
echo "pmc dense" >> $O/progress
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B --response dense > $O/pmc_fetch_dense.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B --response dense > $O/pmc_write_dense.log 2>&1 || exit 1
(cd $R && LDSIM_TRAFFIC_KEY=module0_dense python3 tools/pmc_traffic.py module0 $O/pmc_fetch $O/pmc_write > $O/traffic_dense.log 2>&1 && cp profiles/r04_traffic.json $O/r04_traffic.json) || exit 1
echo "kernel stats" >> $O/progress; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1 || exit 1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_module0.csv
echo "kernel stats 2x2 light" >> $O/progress; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 $R/bench.py --config 2x2_no_modvar --light on --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/kt2.log 2>&1 || exit 1
cp $(find $O/kt2 -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_2x2_light.csv
echo "sq" >> $O/progress; timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/split_profile.py module0 > $O/sq.log 2>&1 || exit 1
(cd $R && python3 tools/pmc_sq.py $O/sq gtables gcorr pixel_adc > $O/r04_sq_counters_module0.txt)
rm -rf $O/pmc_fetch $O/pmc_write $O/kt $O/kt2 $O/sq
cd $R
echo "bench module0" >> $O/progress; python3 bench.py > $O/bench_module0.log 2>&1 || exit 1
echo "bench 2x2" >> $O/progress; python3 bench.py --config 2x2_no_modvar --no-cpu-baseline > $O/bench_2x2.log 2>&1 || exit 1
echo "bench 2x2 light" >> $O/progress; python3 bench.py --config 2x2_no_modvar --light on --no-cpu-baseline --no-extras > $O/bench_2x2_light.log 2>&1 || exit 1
echo "bench ndlar" >> $O/progress; python3 bench.py --config ndlar --light on --no-cpu-baseline > $O/bench_ndlar_light.log 2>&1 || exit 1
echo "phases" >> $O/progress
timeout -k 10 200 python3 tools/gform_phases.py module0 survey 50000 tables > $O/phases_tables.log 2>&1
timeout -k 10 200 python3 tools/gform_phases.py module0 survey 50000 corr > $O/phases_corr.log 2>&1
timeout -k 10 200 python3 tools/gform_phases.py module0 survey 50000 adc > $O/phases_adc.log 2>&1
echo "done" >> $O/progress
for f in module0 2x2 2x2_light ndlar_light; do tail -1 $O/bench_$f.log | cut -c1-260; done
