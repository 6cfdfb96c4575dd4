set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_sweep10k_prof -- python3 $R/bench.py --sweep --steps 2 --warmup 1 > $R/gpurun_out/r03_sweep10k_prof.log 2>&1
cat $R/gpurun_out/r03_sweep10k_prof/*/*kernel_stats.csv | cut -c1-260
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_sweep1k_prof -- python3 $R/tools/sweep_probe.py 1024 262144 > $R/gpurun_out/r03_sweep1k_prof.log 2>&1
cat $R/gpurun_out/r03_sweep1k_prof/*/*kernel_stats.csv | cut -c1-260
cd $R && python -m pytest tests/test_gpu_round2.py -q -k "collapse_run or low_vol" 2>&1 | tail -3
