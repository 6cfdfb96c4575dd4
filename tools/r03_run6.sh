set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R && python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -k "sweep or config4 or mfma or tile or shard or many_portfolios" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_sweep10k_prof2 -- python3 $R/bench.py --sweep --steps 2 --warmup 1 > $R/gpurun_out/r03_sweep10k_prof2.log 2>&1
cat $R/gpurun_out/r03_sweep10k_prof2/*/*kernel_stats.csv | cut -c1-200
cd $R; python tools/sweep_probe.py 1024 262144; python tools/sweep_probe.py 256 262144; python tools/sweep_probe.py 2500 131072
