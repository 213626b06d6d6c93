#!/bin/bash
# Runs on the GPU box: for each config, alternates between the library builds under _ab/<name>/ and prints one line each.
# usage: scripts/gpu_ab_cfgs.sh "C2 C3" <name> <name> ...
cfgs=$1; shift
mkdir -p gpurun_out
for c in $cfgs; do for rep in 1 2; do for v in "$@"; do
    cp _ab/$v/libl3d_hip.so laser_3d_reconstruction_amd/libl3d_hip.so || exit 1
    timeout -k 10 200 python bench.py --config $c --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline > gpurun_out/ab_${c}_$v.json 2> gpurun_out/ab_${c}_$v.err || { tail -5 gpurun_out/ab_${c}_$v.err; exit 1; }
    python - $c $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}_{sys.argv[2]}.json"))
print(sys.argv[1], sys.argv[2], "fps %.1f ms %.2f ok %s" % (d["value"], d["ms_per_step"], d["verified"]), {k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items() if "vertical" in k or "horizontal" in k})
PY
done; done; done
