#!/bin/bash
# Runs on the GPU box: like gpu_ab.sh, with one environment assignment applied to every run (e.g. L3D_SGBM_SPLIT=0).
# usage: scripts/gpu_ab_env.sh VAR=value <name> <name> ... [-- bench.py arguments]
export "$1"; shift
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in "${names[@]}"; do
    cp _ab/$v/libl3d_hip.so laser_3d_reconstruction_amd/libl3d_hip.so || exit 1
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
    python - "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(sys.argv[1], "fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3))
PY
done
done
