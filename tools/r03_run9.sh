R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tools/tail_ab.py --rounds 8 --steps 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_tail_ab.txt; cat gpurun_out/r03_tail_ab.txt
python -m pytest tests -m gpu -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
B="--steps 60 --warmup 10 --no-cpu-baseline --sustain 0"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_tail_cu8 -- python3 $R/bench.py $B --cu-reserve 8 > $R/gpurun_out/r03_tail_cu8.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_tail_ls2 -- python3 $R/bench.py $B --logical-shards 2 > $R/gpurun_out/r03_tail_ls2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_tail_ls2_cu8 -- python3 $R/bench.py $B --logical-shards 2 --cu-reserve 8 > $R/gpurun_out/r03_tail_ls2_cu8.log 2>&1
cd $R
for d in r03_tail_cu8 r03_tail_ls2 r03_tail_ls2_cu8; do
  S=1; case $d in *ls2*) S=2;; esac
  python tools/tail_report.py gpurun_out/$d/*/*_kernel_trace.csv --shards $S > gpurun_out/$d.txt 2>&1; cat gpurun_out/$d.txt
  rm -rf gpurun_out/$d
done
