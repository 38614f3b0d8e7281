#!/bin/bash
# HBM bytes actually moved by the bare write-heavy patterns of profiles/ubench/streams.hip (28 B/px read, 41 B/px written
# algorithmic), per launch shape: separate FETCH_SIZE / WRITE_SIZE passes.  usage: bash profiles/r04_streams_pmc.sh <outdir-name>
set -o pipefail
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o f --output-format csv -- $R/profiles/ubench/streams write_heavy > /dev/null 2> $O/f.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o w --output-format csv -- $R/profiles/ubench/streams write_heavy > /dev/null 2> $O/w.err && echo write ok
python3 - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
px = 16384 * 16384
out = collections.OrderedDict()
for tag, cname, mul in (("read_B_per_px", "FETCH_SIZE", 2.0), ("write_B_per_px", "WRITE_SIZE", 1.0)):
    f = glob.glob(f"{O}/{'f' if cname == 'FETCH_SIZE' else 'w'}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname and "k_rw" in r["Kernel_Name"]:
            key = r["Kernel_Name"].split("(")[0].replace("void ", "") + f" grid={r.get('Grid_Size', '?')}"
            per.setdefault(key, []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        out.setdefault(k, {})[tag] = round(mul * 1024.0 * (sum(v[1:]) / max(len(v) - 1, 1)) / px, 2)   # the first launch is the warm-up
json.dump(out, open(f"{O}/streams_write_heavy_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/f $O/w
