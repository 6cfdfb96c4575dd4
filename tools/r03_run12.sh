R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tools/kernel_lab.py run base sh3 sh4 --portfolios 256 --paths 262144 --rounds 6 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_lab_sweep256.txt; cat gpurun_out/r03_lab_sweep256.txt
python tools/kernel_lab.py run base sh3 sh4 --portfolios 256 --paths 262144 --rounds 6 --no-stats 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_lab_sweep256.txt; tail -4 gpurun_out/r03_lab_sweep256.txt
