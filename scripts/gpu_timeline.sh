#!/bin/bash
# Runs on the GPU box: kernel trace of a few bench steps for each library under _ab/<name>/, then the launch timeline
# (start offset, duration, queue) of the last step.  usage: scripts/gpu_timeline.sh <name> [<name> ...]
ROOT=$GRAFT_REPO_ROOT
for v in "$@"; do
    cp $ROOT/_ab/$v/libl3d_hip.so $ROOT/laser_3d_reconstruction_amd/libl3d_hip.so || exit 1
    out=$ROOT/gpurun_out/tl_$v
    rm -rf $out; mkdir -p $out
    (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out -o tl -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-verify > $out/bench.json 2> $out/err.txt) || { tail -5 $out/err.txt; exit 1; }
    python3 $ROOT/scripts/timeline.py $out/tl_kernel_trace.csv | tail -40
done
