#!/bin/bash
# HBM traffic of the fused index + projection pass under its two workgroup -> pixel mappings (RSSEG_FUSE_MAP 0: grid-stride,
# 1: contiguous chunks), separate FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 2 --warmup 1`, every dispatch listed
# (the first dispatch of a process writes freshly allocated planes).  usage: bash profiles/r04_fuse_pmc.sh <outdir-name>
set -o pipefail
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
for m in 0 1; do
  export RSSEG_FUSE_MAP=$m
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f$m -o f --output-format csv -- $B > /dev/null 2> $O/f$m.err && echo fetch$m ok
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w$m -o w --output-format csv -- $B > /dev/null 2> $O/w$m.err && echo write$m ok
  python3 - $O $m <<'PY'
import csv, sys, glob
O, m = sys.argv[1], sys.argv[2]
px = 16384 * 16384
for d, cname, mul in (("f", "FETCH_SIZE", 2.0), ("w", "WRITE_SIZE", 1.0)):
    f = glob.glob(f"{O}/{d}{m}/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == cname and "k3_indices_project" in r["Kernel_Name"]]
    print(f"map {m} {cname} B/px per dispatch:", [round(mul * 1024 * v / px, 2) for v in vals])
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == cname and "k4_glcm_quad" in r["Kernel_Name"]]
    print(f"      glcm_quad {cname} B/px per dispatch:", [round(mul * 1024 * v / px, 2) for v in vals])
PY
  rm -rf $O/f$m $O/w$m
done
