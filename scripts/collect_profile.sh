#!/bin/bash
# Runs in the build container after scripts/gpu_profile.sh <cfg>: copies the summaries merged into
# gpurun_out/profile_<cfg>/ to profiles/ under the names of the given tag and derives the traffic tables.
# usage: scripts/collect_profile.sh <tag> <cfg> <batch per launch>
tag=$1; cfg=$2; batch=$3
src=gpurun_out/profile_$cfg
cp $src/stats_split/vm_kernel_stats.csv profiles/${tag}_bench_${cfg}_kernel_stats.csv
cp $src/stats/vm_kernel_stats.csv profiles/${tag}_bench_${cfg}_wholebatch_kernel_stats.csv
grep '^{' $src/bench_split_under_profiler.json | tail -1 > profiles/${tag}_bench_${cfg}_under_profiler.json
grep '^{' $src/bench_under_profiler.json | tail -1 > profiles/${tag}_bench_${cfg}_wholebatch_under_profiler.json
python3 scripts/pmc_to_traffic.py $cfg $src/fetch/vm_counter_collection.csv $src/write/vm_counter_collection.csv $tag $batch
[ -f gpurun_out/${tag}_bench_${cfg}.json ] && grep '^{' gpurun_out/${tag}_bench_${cfg}.json | tail -1 > profiles/${tag}_bench_${cfg}.json
ls -la profiles | grep ${cfg}
