#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the bench command, then separate
# PMC passes for HBM read / write traffic.  Raw output goes to gpurun_out/prof_<tag>/ (scratch); the
# summaries that are judged are copied into profiles/ by tools/summarize_profile.py.
set -u
TAG=${1:-r01}
STEPS=${2:-50}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace_stdout.log 2>&1
echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/pmc_fetch_stdout.log 2>&1
echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/pmc_write_stdout.log 2>&1
echo "pmc write rc=$?"
cd $REPO
find $OUT -name "*.csv" | head -30
du -sh $OUT
python tools/summarize_profile.py $OUT $TAG
tail -2 $OUT/trace_stdout.log
# keep only the summaries (the raw traces exceed gpurun's 64 MiB copy-back cap)
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
