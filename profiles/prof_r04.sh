#!/bin/bash
# usage (on the GPU box): bash profiles/prof_r04.sh <outdir-name>     -> gpurun_out/<outdir-name>/...
# Kernel statistics, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes), SQ and MFMA counters of the bench command,
# then the plain bench lines.  Counter passes carry --kernel-trace only (no other trace domain).
set -o pipefail
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/c3_stats -o c3 --output-format csv -- $B --steps 3 --warmup 1 > $O/c3_stats.json 2> $O/c3_stats.err && echo c3_stats ok
rocprofv3 --kernel-trace --stats -d $O/c3h_stats -o c3h --output-format csv -- $B --steps 2 --warmup 1 --data hard > $O/c3h_stats.json 2> $O/c3h_stats.err && echo c3h_stats ok
rocprofv3 --kernel-trace --stats -d $O/c5_stats -o c5 --output-format csv -- $B --config c5 --steps 2 --warmup 1 > $O/c5_stats.json 2> $O/c5_stats.err && echo c5_stats ok
for cfg in c3 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${cfg}_fetch -o f --output-format csv -- $B --config $cfg --steps 1 --warmup 0 > /dev/null 2> $O/${cfg}_fetch.err && echo ${cfg}_fetch ok
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${cfg}_write -o w --output-format csv -- $B --config $cfg --steps 1 --warmup 0 > /dev/null 2> $O/${cfg}_write.err && echo ${cfg}_write ok
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/${cfg}_sq -o s --output-format csv -- $B --config $cfg --steps 1 --warmup 0 > /dev/null 2> $O/${cfg}_sq.err && echo ${cfg}_sq ok
done
for cfg in c3 c5; do
  f=$(find $O/${cfg}_fetch -name "*counter_collection.csv" | head -1); w=$(find $O/${cfg}_write -name "*counter_collection.csv" | head -1)
  python3 $R/profiles/summarize_pmc.py $f $w 16384 16384 $O/r04_${cfg}_pmc_traffic && echo ${cfg}_traffic ok
  s=$(find $O/${cfg}_sq -name "*counter_collection.csv" | head -1); k=$(find $O/${cfg}_sq -name "*kernel_trace.csv" | head -1)
  python3 $R/profiles/summarize_sq.py $s $k $O/r04_${cfg}_pmc_sq.md && echo ${cfg}_sq_summary ok
done
cp $(find $O/c3_stats -name "*kernel_stats.csv" | head -1) $O/r04_c3_16384_kernel_stats.csv
cp $(find $O/c3h_stats -name "*kernel_stats.csv" | head -1) $O/r04_c3_hard_16384_kernel_stats.csv
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) $O/r04_c5_16384_kernel_stats.csv
rm -rf $O/c3_stats $O/c3h_stats $O/c5_stats $O/*_fetch $O/*_write $O/*_sq
python3 $R/bench.py > $O/r04_bench_c3.json 2> $O/bench_c3.err && echo bench_c3 ok
python3 $R/bench.py --data hard --no-extras > $O/r04_bench_c3_hard.json 2> $O/bench_c3h.err && echo bench_c3_hard ok
python3 $R/bench.py --config c5 > $O/r04_bench_c5.json 2> $O/bench_c5.err && echo bench_c5 ok
python3 $R/bench.py --config c2 > $O/r04_bench_c2.json 2> $O/bench_c2.err && echo bench_c2 ok
# every collective of a step through a ONE-rank RCCL communicator: driven by the library itself (native), through the
# torch.distributed callback, and the same call without any communication path
python3 $R/bench.py --no-extras --no-cpu-baseline --rccl-single --comm native > $O/r04_bench_c3_rccl_single_rank.json 2> $O/rccl_native.err && echo rccl_native ok
python3 $R/bench.py --no-extras --no-cpu-baseline --rccl-single --comm torch > $O/r04_bench_c3_rccl_single_rank_torch_hook.json 2> $O/rccl_torch.err && echo rccl_torch ok
python3 $R/bench.py --no-extras --no-cpu-baseline > $O/r04_bench_c3_no_comm_same_call.json 2> $O/nocomm.err && echo nocomm ok
python3 $R/bench.py --no-extras --no-cpu-baseline --config c5 --rccl-single --comm native > $O/r04_bench_c5_rccl_single_rank.json 2> $O/rccl_c5.err && echo rccl_c5 ok
# one rank of eight alone: a 2048 x 16384 stripe (profiles/ab_stripe.py), and its kernel timeline
python3 $R/profiles/ab_stripe.py 2048 - easy > $O/r04_stripe_2048rows.txt 2> $O/stripe.err && echo stripe ok
rocprofv3 --kernel-trace -d $O/tl -o tl --output-format csv -- python3 $R/profiles/ab_stripe.py 2048 - easy > /dev/null 2> $O/tl.err && python3 $R/profiles/trace_gaps.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/r04_timeline_2048rows.txt 2>> $O/tl.err && echo timeline_stripe ok
rocprofv3 --kernel-trace -d $O/tl2 -o tl --output-format csv -- python3 $R/profiles/ab_stripe.py 16384 - easy > /dev/null 2> $O/tl2.err && python3 $R/profiles/trace_gaps.py $(find $O/tl2 -name "*kernel_trace.csv" | head -1) > $O/r04_timeline_16384.txt 2>> $O/tl2.err && echo timeline_full ok
rm -rf $O/tl $O/tl2
ls $O
