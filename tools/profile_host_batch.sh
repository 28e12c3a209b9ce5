#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel + memory-copy trace of swmi_score_batch at 1M and 4M host pairs (no PMC: copy traces and
# counters are never combined).  Output: gpurun_out/<tag>_host_batch_trace.txt
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_${TAG}_host
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t -o host --output-format csv -- python3 $ROOT/tools/host_batch_trace.py > $OUT/run.log 2>&1 || echo "trace failed"
cat $OUT/run.log | grep "^n " 
K=$(find $OUT/t -name "*kernel_trace.csv" | head -1); M=$(find $OUT/t -name "*memory_copy_trace.csv" | head -1)
head -2 $M
( cat $OUT/run.log | grep "^n "; python3 $ROOT/tools/summarize_host_trace.py $K $M ) > $ROOT/gpurun_out/${TAG}_host_batch_trace.txt
cat $ROOT/gpurun_out/${TAG}_host_batch_trace.txt | head -60
