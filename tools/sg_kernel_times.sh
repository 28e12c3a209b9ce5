#!/bin/bash
# Run ON THE GPU BOX (via gpurun): per-kernel durations of the semi-global aligner at 65536 alignments (bench.py --mode
# semiglobal) from rocprofv3 --kernel-trace --stats, after a parity run of tests/test_semiglobal.py.  Output: gpurun_out/<tag>_sg_kernel_stats.csv
set -u
TAG=${1:-sg}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $ROOT && timeout -k 10 300 python -m pytest tests/test_semiglobal.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/sg -o sg --output-format csv -- python3 $ROOT/bench.py --mode semiglobal --no-cpu-baseline ${2:-} > $OUT/sg.log 2>&1 || echo "trace failed"
f=$(find $OUT/sg -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_sg_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if "sg_" in r[0]:
        print("%-28s calls %3s  avg %.3f ms" % (r[0].split("(anonymous namespace)::")[1].split("(")[0][:28], r[1], float(r[3]) / 1e6))
PY
tail -1 $OUT/sg.log | cut -c1-200
