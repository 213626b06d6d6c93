#!/bin/bash
# Runs on the GPU box: the two HBM-traffic counter passes only (see gpu_profile.sh), then prints per-kernel bytes.
# usage: scripts/gpu_traffic.sh <tag> [bench.py arguments...]
tag=$1; shift
ROOT=$GRAFT_REPO_ROOT
out=$ROOT/gpurun_out/traffic_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o vm -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $out/fetch.err || { tail -5 $out/fetch.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o vm -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $out/write.err || { tail -5 $out/write.err; exit 1; }
python3 $ROOT/scripts/pmc_to_traffic.py --print-only $out/fetch/vm_counter_collection.csv $out/write/vm_counter_collection.csv
