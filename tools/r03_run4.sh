set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tools/host_call_probe.py --split > gpurun_out/r03_hostcall_a.txt 2>&1; cat gpurun_out/r03_hostcall_a.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_hc_trace -- python3 $R/tools/host_call_probe.py --split > $R/gpurun_out/r03_hc_trace.log 2>&1
cat $R/gpurun_out/r03_hc_trace/*/*kernel_stats.csv | head -30
cd $R
python bench.py --sweep --steps 2 --warmup 1 > gpurun_out/r03_sweep10k_a.json 2>gpurun_out/r03_sweep10k_a.err; cat gpurun_out/r03_sweep10k_a.json
python tools/sweep_probe.py 1024 262144 > gpurun_out/r03_sweep1k_a.txt 2>&1; cat gpurun_out/r03_sweep1k_a.txt
