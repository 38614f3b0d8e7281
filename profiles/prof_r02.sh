#!/bin/bash
# usage (on the GPU box): bash profiles/prof_r02.sh <outdir-name>
set -o pipefail
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/c3_stats -o c3 --output-format csv -- $B --steps 3 --warmup 1 > $O/c3_stats.json 2> $O/c3_stats.err && echo c3_stats ok
rocprofv3 --kernel-trace --stats -d $O/c5_stats -o c5 --output-format csv -- $B --config c5 --steps 2 --warmup 1 > $O/c5_stats.json 2> $O/c5_stats.err && echo c5_stats ok
for cfg in c3 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${cfg}_fetch -o f --output-format csv -- $B --config $cfg --steps 1 --warmup 0 > /dev/null 2> $O/${cfg}_fetch.err && echo ${cfg}_fetch ok
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${cfg}_write -o w --output-format csv -- $B --config $cfg --steps 1 --warmup 0 > /dev/null 2> $O/${cfg}_write.err && echo ${cfg}_write ok
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $O/${cfg}_sq -o s --output-format csv -- $B --config $cfg --steps 1 --warmup 0 > /dev/null 2> $O/${cfg}_sq.err && echo ${cfg}_sq ok
done
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err && echo bench_c3 ok
python3 $R/bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err && echo bench_c5 ok
python3 $R/bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err && echo bench_c2 ok
ls $O
