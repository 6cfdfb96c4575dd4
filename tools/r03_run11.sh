R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_parity.py -q -k "sweep or mfma" 2>&1 | tail -5
for k in 256 200 1250 2500; do python tools/sweep_probe.py $k 262144 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03_sweep_shapes.txt; cat gpurun_out/r03_sweep_shapes.txt
python tools/kernel_lab.py run base w4big nokeys w4nokeys --assets 64 --steps 1260 --paths 1000000 --rounds 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_lab_n64.txt; cat gpurun_out/r03_lab_n64.txt
