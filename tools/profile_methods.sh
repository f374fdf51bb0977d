#!/bin/bash
# PMC passes (HBM read / write traffic) for the Adams kernels in situ: tools/methods_bench.py restricted to the two
# Adams methods.  Same counters, passes and corrections as tools/profile_gpu.sh; summaries by tools/summarize_profile.py.
set -u
TAG=${1:-r01h_methods}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp TDEQ_BENCH_STEPS=16
cd /tmp
CMD="python $REPO/tools/methods_bench.py explicit_adams implicit_adams"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch_stdout.log 2>&1
echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write_stdout.log 2>&1
echo "pmc write rc=$?"
cd $REPO
python tools/summarize_profile.py $OUT $TAG | tail -30
rm -rf $OUT/pmc_fetch $OUT/pmc_write
