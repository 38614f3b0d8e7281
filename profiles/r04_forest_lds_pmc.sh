#!/bin/bash
# LDS counters of the forest kernel (config 5): bank conflicts against active cycles.  bash profiles/r04_forest_lds_pmc.sh <outdir>
set -o pipefail
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras --config c5 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d $O/lds -o l --output-format csv -- $B > /dev/null 2> $O/lds.err && echo lds ok
f=$(find $O/lds -name "*counter_collection.csv" | head -1)
python3 - "$f" > $O/r04_forest_lds_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "forest" in k:
        acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, dict(v))
PY
cat $O/r04_forest_lds_pmc.txt
rm -rf $O/lds
