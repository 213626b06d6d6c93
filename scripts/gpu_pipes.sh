#!/bin/bash
# Runs on the GPU box: bench.py for 1..4 pipelines in flight, one summary line each.  usage: scripts/gpu_pipes.sh [bench.py arguments]
mkdir -p gpurun_out
for p in 1 2 3 4; do
    timeout -k 10 200 python bench.py --steps 24 --warmup 4 --repeats 3 --no-cpu-baseline --no-verify --pipelines $p "$@" > gpurun_out/pipes_$p.json 2> gpurun_out/pipes_$p.err || { tail -5 gpurun_out/pipes_$p.err; exit 1; }
    python - $p <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/pipes_{sys.argv[1]}.json"))
print("pipelines", sys.argv[1], "fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "min/max", round(d["timing"]["value_min"]), round(d["timing"]["value_max"]))
PY
done
