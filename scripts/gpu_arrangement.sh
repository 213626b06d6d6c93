#!/bin/bash
# Runs on the GPU box: step time of one configuration for pipelines in flight x half-batch split.
# usage: scripts/gpu_arrangement.sh <config> "<pipelines ...>" "<split values ...>"
cfg=$1; pipes=$2; splits=$3
for s in $splits; do for p in $pipes; do
    L3D_SGBM_SPLIT=$s timeout -k 10 200 python bench.py --config $cfg --pipelines $p --steps 12 --warmup 3 --repeats 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'split', $s, 'pipelines', $p, round(d['value'],1), round(d['ms_per_step'],3), d['verified'])" || exit 1
done; done
