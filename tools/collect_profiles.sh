#!/bin/bash
# Copy the summaries of one tools/profile_round.sh run into profiles/<round>/ (tracked).
# usage: tools/collect_profiles.sh <tag of gpurun_out/prof_<tag>> <round dir, e.g. r02> <workload> <version, e.g. v8>
set -e
tag=$1; rnd=$2; wl=$3; ver=$4
src=gpurun_out/prof_$tag; dst=profiles/$rnd
mkdir -p $dst
cp $(ls $src/trace/*/*kernel_stats.csv | head -1) $dst/bench_${wl}_kernel_stats_$ver.csv
python tools/prof_summary.py $src/trace > $dst/bench_${wl}_kernel_by_grid_$ver.txt
cp $src/bench.json $dst/bench_${wl}_$ver.json
cp $src/bench_under_rocprof.json $dst/bench_${wl}_${ver}_under_rocprof.json
python tools/pmc_traffic.py $src $dst/pmc_traffic_${wl}.json > /dev/null
ls -la $dst
