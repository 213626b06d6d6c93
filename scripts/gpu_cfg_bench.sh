#!/bin/bash
# Runs on the GPU box: bench.py for the given configs and pipeline counts, one summary line each.
# usage: scripts/gpu_cfg_bench.sh "C3 C5" "1 2" [extra bench.py arguments...]
cfgs=$1; pipes=$2; shift 2
mkdir -p gpurun_out
for c in $cfgs; do for p in $pipes; do
    timeout -k 10 280 python bench.py --config $c --pipelines $p --steps 6 --warmup 2 --repeats 3 --no-cpu-baseline "$@" > gpurun_out/r2_${c}_p$p.json 2> gpurun_out/r2_${c}_p$p.err || { tail -3 gpurun_out/r2_${c}_p$p.err; exit 1; }
    python - $c $p <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r2_{sys.argv[1]}_p{sys.argv[2]}.json"))
print(sys.argv[1], "pipelines", sys.argv[2], "fps %.1f ms/step %.2f verified %s" % (d["value"], d["ms_per_step"], d["verified"]))
print("  stages", {k: round(v, 2) for k, v in d["stage_ms_per_step_serialised"].items()})
print("  kernels", {k: round(v, 2) for k, v in d["kernel_ms_per_launch"].items()})
PY
done; done
