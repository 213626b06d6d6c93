#!/bin/bash
# Runs on the GPU box: half-batch split on / off against pipelines in flight, alternating, twice over.
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "1 2" "0 2" "0 3" "0 4" "1 1"; do
    set -- $cfg
    L3D_SGBM_SPLIT=$1 timeout -k 10 200 python bench.py --steps 24 --warmup 4 --repeats 3 --no-cpu-baseline --no-verify --pipelines $2 > gpurun_out/sp_$1_$2.json 2> gpurun_out/sp_$1_$2.err || { tail -5 gpurun_out/sp_$1_$2.err; exit 1; }
    python - $1 $2 <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/sp_{sys.argv[1]}_{sys.argv[2]}.json"))
print("split", sys.argv[1], "pipelines", sys.argv[2], "fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3))
PY
done
done
