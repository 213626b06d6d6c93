#!/bin/bash
# Runs on the GPU box: alternates bench.py runs with and without an environment switch, one line each.
# usage: scripts/gpu_ab_env2.sh VAR "cfg cfg ..." [bench.py arguments...]
var=$1; cfgs=$2; shift 2
mkdir -p gpurun_out
for c in $cfgs; do for rep in 1 2; do for v in "" 1; do
    if [ -n "$v" ]; then export $var=1; else unset $var; fi
    timeout -k 10 200 python bench.py --config $c --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline "$@" > gpurun_out/ab2.json 2> gpurun_out/ab2.err || { tail -5 gpurun_out/ab2.err; exit 1; }
    python - $c "$var=$v" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab2.json") if l.startswith("{")][-1])
print(sys.argv[1], sys.argv[2], "fps %.1f ms %.3f ok %s" % (d["value"], d["ms_per_step"], d["verified"]), {k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items() if "vertical" in k or "horizontal" in k})
PY
done; done; done
