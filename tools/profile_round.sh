#!/bin/bash
# Kernel-trace/stats profiles of the shapes DESIGN.md quotes (run on the GPU box):  tools/profile_round.sh <tag>
# rocprofv3 with --kernel-trace --stats only (no PMC here; counters are tools/pmc_passes.sh).  Every configuration is a fresh
# process (stream -> hardware-queue placement is per process).
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats --output-format csv"
B="--no-cpu-baseline --sustain 0"
$P -d $O/bench     -- python3 $R/bench.py --steps 40 --warmup 5 $B > $O/bench.log 2>&1
$P -d $O/serial    -- python3 $R/bench.py --serial --steps 40 --warmup 5 $B > $O/serial.json 2> $O/serial.err
$P -d $O/ls2       -- python3 $R/bench.py --steps 40 --warmup 5 $B --logical-shards 2 > $O/ls2.log 2>&1
$P -d $O/shard125  -- python3 $R/bench.py --paths-per-gpu 12500000 --steps 4 --warmup 1 $B > $O/shard125.json 2> $O/shard125.err
$P -d $O/sweep10k  -- python3 $R/bench.py --sweep --steps 2 --warmup 1 > $O/sweep10k.json 2> $O/sweep10k.err
$P -d $O/sweep2500 -- python3 $R/bench.py --sweep --sweep-portfolios 2500 --sweep-paths 262144 --steps 2 --warmup 1 > $O/sweep2500.json 2> $O/sweep2500.err
$P -d $O/sweep1k   -- python3 $R/tools/sweep_probe.py 1024 262144 > $O/sweep1k.txt 2>&1
$P -d $O/sweep256  -- python3 $R/tools/sweep_probe.py 256 262144 > $O/sweep256.txt 2>&1
$P -d $O/n64       -- python3 $R/tools/profile_paths.py --assets 64 --steps 1260 --paths 10000000 --launches 2 > $O/n64.txt 2>&1

cd $R
for d in bench serial ls2 shard125 sweep10k sweep2500 sweep1k sweep256 n64; do cp $O/$d/*/*_kernel_stats.csv $O/${d}_kernel_stats.csv; done
python3 tools/trace_gaps.py $(ls $O/bench/*/*_kernel_trace.csv | head -1) 0 60 --overlap > $O/overlap.txt 2>&1 || true
python3 tools/tail_report.py $(ls $O/bench/*/*_kernel_trace.csv | head -1) > $O/tail_1shard.txt 2>&1 || true
python3 tools/tail_report.py $(ls $O/ls2/*/*_kernel_trace.csv | head -1) --shards 2 > $O/tail_2shards.txt 2>&1 || true
for d in bench serial ls2 shard125 sweep10k sweep2500 sweep1k sweep256 n64; do rm -rf $O/$d; done
# un-profiled figures, each a fresh process
python3 tools/host_call_probe.py > $O/hostcall.txt 2>&1
python3 tools/host_call_probe.py --split >> $O/hostcall.txt 2>&1
python3 bench.py --config3 --steps 2 --warmup 1 > $O/config3.json 2>/dev/null
python3 bench.py --steps 40 --warmup 5 $B > $O/bench_1shard.json 2>/dev/null
python3 bench.py --steps 40 --warmup 5 $B --logical-shards 2 > $O/bench_2shards.json 2>/dev/null
python3 bench.py --steps 40 --warmup 5 $B --logical-shards 2 --skew 1 --stats-streams 1 > $O/bench_2shards_skew.json 2>/dev/null
python3 tools/config4_full.py > $O/config4_full.txt 2>&1 || true
python3 tools/sweep_hist_probe.py 2>&1 | grep -v amdgpu.ids > $O/sweep_hist.txt || true
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep_hist -- python3 $R/tools/sweep_hist_probe.py 4096 16 2500 > /dev/null 2>&1; cp $O/sweep_hist/*/*_kernel_stats.csv $O/sweep_hist_kernel_stats.csv; rm -rf $O/sweep_hist) || true
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -3 $O/hostcall.txt; tail -2 $O/overlap.txt; cat $O/tail_1shard.txt | head -4; cat $O/tail_2shards.txt | head -4
