#!/bin/bash
# One round's profile set for bench.py (config 3): kernel trace + stats, then the PMC passes (each in
# its own run, as gpurun requires), then the calibration passes.  usage: tools/profile_round.sh <tag> [c3|c4|c5]
# Writes under gpurun_out/prof_<tag>/; summarise with tools/prof_summary.py and tools/pmc_traffic.py.
set -e
tag=${1:-v}
wl=${2:-c3}
root=$GRAFT_REPO_ROOT
[ -z "$root" ] && root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --fresh 0 --workload $wl"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/bench_under_rocprof.json 2> $out/trace.log
echo "trace done" >> $out/progress
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $BENCH > $out/fetch.log 2>&1
echo "fetch done" >> $out/progress
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $BENCH > $out/write.log 2>&1
echo "write done" >> $out/progress
timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- python3 $root/tools/pmc_calibrate.py > $out/cal_fetch.log 2>&1
timeout -s KILL 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- python3 $root/tools/pmc_calibrate.py > $out/cal_write.log 2>&1
echo "cal done" >> $out/progress
timeout -k 10 300 python3 $root/bench.py --steps 30 --warmup 8 --workload $wl > $out/bench.json 2> $out/bench.err
echo "bench done" >> $out/progress
# keep only what the summaries read (the merge back is capped at 64 MiB)
find $out -name "*.csv" ! -name "*kernel_trace.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
python3 $root/tools/pmc_traffic.py $out $out/pmc_traffic_$wl.json > /dev/null 2>&1 || true
python3 $root/tools/prof_summary.py $out/trace > $out/kernel_by_grid.txt 2>&1 || true
find $out -name "*counter_collection.csv" -delete
du -sh $out
