set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -m gpu -q > gpurun_out/r03_gputest3.log 2>&1; tail -5 gpurun_out/r03_gputest3.log
python bench.py --steps 20 --warmup 5 --sustain 3 > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; tail -c 600 gpurun_out/r03_bench_a.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --logical-shards 2 > gpurun_out/r03_bench_ls2.json 2> gpurun_out/r03_bench_ls2.err; head -c 400 gpurun_out/r03_bench_ls2.json
tools/microbench/mfma_coexec > gpurun_out/r03_mfma_coexec.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r03_coexec_pmc -- $R/tools/microbench/mfma_coexec > $R/gpurun_out/r03_coexec_pmc.log 2>&1
ls -R $R/gpurun_out/r03_coexec_pmc | head
