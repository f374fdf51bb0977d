#!/bin/bash
# r06: are the re-reads of k_0..k_4 by successive stage rows of a SHARD-sized trial step L2 hits or
# fabric traffic?  Run on the GPU box (through gpurun):
#   tools/shard_l2.sh <tag> "<command>"        e.g.  tools/shard_l2.sh cfg2_shard_graph "python $PWD/tools/run_config.py cfg2_shard 20"
# (the command runs from /tmp: give absolute paths)
# Passes (PMC never combined with tracing domains other than --kernel-trace): kernel trace + stats; TCC_HIT_sum + TCC_MISS_sum;
# FETCH_SIZE; WRITE_SIZE; TCC_REQ_sum + TCC_READ_sum.  tools/shard_l2_summary.py condenses them per kernel.
set -u
TAG=$1
CMD=$2
REPO=$(pwd)
OUT=$REPO/gpurun_out/l2_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace_stdout.log 2>&1
echo "trace rc=$?"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_hit -o hit -- $CMD > $OUT/pmc_hit_stdout.log 2>&1
echo "pmc hit/miss rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch_stdout.log 2>&1
echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write_stdout.log 2>&1
echo "pmc write rc=$?"
timeout 600 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $OUT/pmc_req -o req -- $CMD > $OUT/pmc_req_stdout.log 2>&1
echo "pmc req rc=$?"
cd $REPO
python tools/shard_l2_summary.py $OUT $TAG > $OUT/${TAG}_l2.json
tail -2 $OUT/trace_stdout.log
rm -rf $OUT/trace $OUT/pmc_hit $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_req
