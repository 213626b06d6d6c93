#!/bin/bash
# usage: scripts/dev_pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]; prints per-kernel sums for the two scan kernels
# (and for the kernels whose name contains $PMC_MATCH, if set)
tag=$1; shift
export SWEEP_BATCHES=${SWEEP_BATCHES:-64} SWEEP_LPP=${SWEEP_LPP:-0}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
n=0
for pass in "$@"; do
  n=$((n+1))
  out=$ROOT/gpurun_out/pmc_${tag}_$n
  rm -rf $out
  timeout -s KILL 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out -o vm -- python3 $ROOT/scripts/dev_sweep.py $SWEEP_ARGS > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("$ROOT/gpurun_out/pmc_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "K2" if "k_cost_vertical" in k else "K3" if "k_horizontal" in k else None
        extra = "$PMC_MATCH"   # optional: substring of another kernel's name to report
        if name is None and extra and extra in k: name = extra
        if name:
            agg[(name, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
for k in sorted(agg): print(k[0], k[1], "%.4g" % (agg[k] / cnt[k]), "per dispatch over", cnt[k])
PY
