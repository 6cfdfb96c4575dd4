#!/bin/bash
# Kernel-trace/stats profiles of the shapes DESIGN.md quotes (run on the GPU box):  tools/profile_round.sh <tag>
# rocprofv3 with --kernel-trace --stats only (no PMC here; counters are tools/pmc_passes.sh).
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats --output-format csv"
$P -d $O/bench     -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
$P -d $O/shard125  -- python3 $R/bench.py --paths-per-gpu 12500000 --steps 4 --warmup 1 --no-cpu-baseline > $O/shard125.log 2>&1
$P -d $O/sweep10k  -- python3 $R/bench.py --sweep --steps 2 --warmup 1 > $O/sweep10k.log 2>&1
$P -d $O/sweep1k   -- python3 $R/tools/sweep_probe.py 1024 262144 > $O/sweep1k.log 2>&1
$P -d $O/n64       -- python3 $R/tools/profile_paths.py --assets 64 --steps 1260 --paths 10000000 --launches 2 > $O/n64.log 2>&1

cd $R
python3 tools/trace_gaps.py $(ls $O/bench/*/*_kernel_trace.csv | head -1) 0 60 --overlap > $O/overlap.txt 2>&1 || true
python3 tools/host_call_probe.py > $O/hostcall.txt 2>&1
python3 tools/host_call_probe.py --split >> $O/hostcall.txt 2>&1
python3 bench.py --config3 --steps 2 --warmup 1 > $O/config3.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -3 $O/hostcall.txt; tail -2 $O/overlap.txt
