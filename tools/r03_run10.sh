R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tools/tail_ab.py --rounds 8 --steps 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_tail_ab.txt; cat gpurun_out/r03_tail_ab.txt
