set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for v in debug prod; do
  cp larnd-sim_amd/larndsim_amd/variants/lib_$v.so larnd-sim_amd/larndsim_amd/libldsim_hip.so
  for a in "module0 survey" "ndlar survey"; do
    timeout -k 10 200 python tools/gform_phases.py $a 50000 share > gpurun_out/ab_tmp.log 2>&1 || exit 1
    echo "$v: $(grep 'dbg  0' gpurun_out/ab_tmp.log | tail -1 | cut -c1-110)"
  done
done
done
