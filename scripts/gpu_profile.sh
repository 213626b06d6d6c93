#!/bin/bash
# Runs on the GPU box: kernel-trace statistics and the two HBM-traffic counter passes of a short
# bench.py run, leaving the summaries under gpurun_out/profile_<tag>/ for copying into profiles/.
# usage: scripts/gpu_profile.sh <tag> [bench.py arguments...]
tag=$1; shift
ROOT=$GRAFT_REPO_ROOT
out=$ROOT/gpurun_out/profile_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify"
# the default command (a batch as two halves on two streams: kernels overlap, their durations stretch) ...
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_split -o vm -- python3 $ROOT/bench.py $COMMON "$@" > $out/bench_split_under_profiler.json 2> $out/stats_split.err || { tail -5 $out/stats_split.err; exit 1; }
# ... and with whole-batch launches on one stream and one pipeline in flight, so that no two kernels overlap: the form
# the per-kernel event timing of bench.py uses, and the summary whose averages are comparable with it
export L3D_SGBM_SPLIT=0
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o vm -- python3 $ROOT/bench.py $COMMON --pipelines 1 "$@" > $out/bench_under_profiler.json 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o vm -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-verify "$@" > /dev/null 2> $out/fetch.err || { tail -5 $out/fetch.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o vm -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-verify "$@" > /dev/null 2> $out/write.err || { tail -5 $out/write.err; exit 1; }
# keep what is small: the statistics tables and the counter rows; the kernel traces themselves stay on the box
rm -f $out/*/vm_kernel_trace.csv $out/*/*agent_info.csv
ls $out/stats $out/fetch $out/write
tail -1 $out/bench_under_profiler.json | cut -c1-300
