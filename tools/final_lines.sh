set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fin
mkdir -p $O
cd $R
echo module0 > $O/progress; python3 bench.py > $O/bench_module0.log 2>&1 || exit 1
echo 2x2 >> $O/progress; python3 bench.py --config 2x2_no_modvar --no-cpu-baseline > $O/bench_2x2.log 2>&1 || exit 1
echo 2x2l >> $O/progress; python3 bench.py --config 2x2_no_modvar --light on --no-cpu-baseline --no-extras > $O/bench_2x2_light.log 2>&1 || exit 1
echo nd >> $O/progress; python3 bench.py --config ndlar --light on --no-cpu-baseline > $O/bench_ndlar_light.log 2>&1 || exit 1
echo 3 >> $O/progress; python3 bench.py --baseline-config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_2x2_1M.log 2>&1 || exit 1
echo 5 >> $O/progress; python3 bench.py --baseline-config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_ndlar_1M.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
echo kt >> $O/progress; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --config ndlar --light on --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1 || exit 1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_ndlar.csv
rm -rf $O/kt
echo done >> $O/progress
for f in module0 2x2 2x2_light ndlar_light 2x2_1M ndlar_1M; do tail -1 $O/bench_$f.log | cut -c1-200; done
