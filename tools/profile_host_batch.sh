#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel + memory-copy trace of a host-buffer entry at 1M and 4M host pairs (no PMC: copy traces
# and counters are never combined).  Usage: tools/profile_host_batch.sh <tag> [pairs|packed|ovm]
# Output: gpurun_out/<tag>_host_batch_trace[_<entry>].txt
set -u
TAG=${1:-r04}
ENTRY=${2:-pairs}
SUF=""; [ "$ENTRY" != "pairs" ] && SUF="_$ENTRY"
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_${TAG}_host$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t -o host --output-format csv -- python3 $ROOT/tools/host_batch_trace.py $ENTRY > $OUT/run.log 2>&1 || echo "trace failed"
K=$(find $OUT/t -name "*kernel_trace.csv" | head -1); M=$(find $OUT/t -name "*memory_copy_trace.csv" | head -1)
( grep "^n " $OUT/run.log; python3 $ROOT/tools/summarize_host_trace.py $K $M ) > $ROOT/gpurun_out/${TAG}_host_batch_trace$SUF.txt
head -70 $ROOT/gpurun_out/${TAG}_host_batch_trace$SUF.txt
