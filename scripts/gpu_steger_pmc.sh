#!/bin/bash
# Runs on the GPU box: hardware counters of the Steger extractor's kernels (scripts/dev_steger.py), one pass per counter set.
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
n=0
for pass in "$@"; do
  n=$((n+1)); out=$ROOT/gpurun_out/steger_pmc_$n; rm -rf $out
  timeout -s KILL 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out -o vm -- python3 $ROOT/scripts/dev_steger.py > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$ROOT/gpurun_out/steger_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "tile" if "k_steger_tile" in k else "mark" if "k_steger_mark" in k else None
        if name: agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = sorted(agg[k]); print(k[0], k[1], "max %.4g median %.4g over %d" % (v[-1], v[len(v)//2], len(v)))
PY
