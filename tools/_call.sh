mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for arr in serial split2_skew; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r02j/tr_$arr -o t -- python $REPO/tools/overlap_probe.py $arr > $REPO/gpurun_out/r02j/$arr.log 2>&1
python $REPO/tools/trace_gaps.py $(find $REPO/gpurun_out/r02j/tr_$arr -name "*kernel_trace.csv" | head -1) 3000 > $REPO/gpurun_out/r02j/overlap_trace_$arr.json
python -c "
import json; o=json.load(open('$REPO/gpurun_out/r02j/overlap_trace_$arr.json')); print('$arr', {k:o[k] for k in ('dispatches','span_us','busy_us','gap_us','us_with_two_or_more_kernels_in_flight')})"
rm -rf $REPO/gpurun_out/r02j/tr_$arr
done
