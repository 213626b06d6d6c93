#!/bin/bash
# Runs on the GPU box: alternates between builds of the library kept under _ab/<name>/libl3d_hip.so
# (same box, same process conditions) and prints the per-kernel times of each run.
# usage: scripts/gpu_ab.sh <name> <name> ... [-- bench.py arguments]
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in "${names[@]}"; do
    cp _ab/$v/libl3d_hip.so laser_3d_reconstruction_amd/libl3d_hip.so || exit 1
    timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
    python - "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(sys.argv[1], "fps", round(d["value"], 1), {k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items()})
PY
done
done
