R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for q in 4 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/tail_ab.py --rounds 6 --steps 40 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03_tail_ab_queues.txt 2>&1
cat gpurun_out/r03_tail_ab_queues.txt
