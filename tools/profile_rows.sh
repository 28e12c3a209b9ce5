#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace statistics of the secondary bench modes (rows N1, N3, N4 and the
# banded-affine extension).  Outputs under gpurun_out/prof_rows/; the *_kernel_stats.csv files are copied to profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_rows
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in semiglobal banded-affine one-vs-many packed; do
  rocprofv3 --kernel-trace --stats -d $OUT/$MODE -o $MODE --output-format csv -- python3 $ROOT/bench.py --mode $MODE --no-cpu-baseline --sg-plain > $OUT/$MODE.log 2>&1 || echo "$MODE failed"
  f=$(find $OUT/$MODE -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${MODE}_kernel_stats.csv
  tail -1 $OUT/$MODE.log | cut -c1-300
done
ls $OUT
