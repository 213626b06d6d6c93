"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per run) into
profiles/traffic.json: HBM bytes per launch for each kernel.

Corrections (MI355X / gfx950): both counters are in KiB; FETCH_SIZE tallies 128-byte fabric reads
at 64 bytes, i.e. reports half the bytes of a coalesced streaming read, so it is doubled (checked
here on k_bgr2gray: 3 B/pixel read -> FETCH_SIZE*2*1024 == 3*W*H*B); WRITE_SIZE is exact.

usage: pmc_to_traffic.py <config> <fetch_counter_collection.csv> <write_counter_collection.csv> [tag] [batch]
       pmc_to_traffic.py --print-only <fetch csv> <write csv>      (prints, leaves profiles/traffic.json alone)

Besides profiles/traffic.json it writes profiles/<tag>_<config>_pmc_per_kernel.csv: per kernel and counter the number
of dispatches and the mean, minimum and maximum raw counter value (KiB), i.e. the rows of the two rocprofv3 counter
files reduced to one line per kernel -- what traffic.json is derived from.
"""
import collections
import csv
import json
import os
import sys

print_only = sys.argv[1] == "--print-only"
cfg, ffile, wfile = sys.argv[1:4]
tag = sys.argv[4] if len(sys.argv) > 4 else "r1"
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 64
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("l3d::", "")
        name = name.split("<")[0]
        acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def raw_table(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return [(k, counter, len(v), sum(v) / len(v), min(v), max(v)) for k, v in sorted(acc.items())]


f, w = per_kernel(ffile), per_kernel(wfile)
out = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    out[k] = {"fetch_bytes": f.get(k, 0.0) * 2 * 1024, "write_bytes": w.get(k, 0.0) * 1024}
    out[k]["hbm_bytes"] = out[k]["fetch_bytes"] + out[k]["write_bytes"]
sp = [k for k in out if k.startswith("k_speckle")]
out["k_speckle_*"] = {"hbm_bytes": sum(out[k]["hbm_bytes"] for k in sp)}
if not print_only:
    path = os.path.join(root, "profiles", "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[cfg] = {k: v["hbm_bytes"] for k, v in out.items()}
    data[cfg + "_batch"] = batch
    data[cfg + "_detail_" + tag] = out
    json.dump(data, open(path, "w"), indent=1)
    with open(os.path.join(root, "profiles", f"{tag}_{cfg}_pmc_per_kernel.csv"), "w") as out_csv:
        out_csv.write("kernel,counter,dispatches,mean_KiB,min_KiB,max_KiB\n")
        for row in raw_table(ffile, "FETCH_SIZE") + raw_table(wfile, "WRITE_SIZE"):
            out_csv.write("\"%s\",%s,%d,%.3f,%.3f,%.3f\n" % row)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes"]):
    print(f"{k:22s} {v['hbm_bytes'] / 1e6:10.1f} MB / launch   fetch {v.get('fetch_bytes', 0) / 1e6:10.1f}  write {v.get('write_bytes', 0) / 1e6:10.1f}")
