#!/bin/bash
# Runs on the GPU box: bench.py once per value of an environment switch, twice over, compact summary.
# usage: scripts/gpu_env_bench.sh VAR=a,b [bench.py arguments]
var=${1%%=*}; vals=${1#*=}; shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in ${vals//,/ }; do
    export $var=$v
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline "$@" > gpurun_out/env_$v.json 2> gpurun_out/env_$v.err || { tail -5 gpurun_out/env_$v.err; exit 1; }
    python - "$var=$v" gpurun_out/env_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items() if not k.startswith("right/")}, "verified", d["verified"])
PY
done
done
