#!/bin/bash
# Run ON THE GPU BOX (via gpurun) after `make -C smith-waterman-simd_amd/csrc ab_walk`: the semi-global walk kernel's duration
# at 65536 alignments as shipped and with its records always in the cache (the decoding alone; the fetch alone is the
# "walk shape" rows of tools/microbench/hbm_stream) -- rocprofv3 --kernel-trace --stats of bench.py --mode semiglobal with SWMI_LIB naming the A/B library.
# (The A/B builds return wrong tracebacks by construction; the bench's parity check is skipped with --no-cpu-baseline.)
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
LIB=$ROOT/smith-waterman-simd_amd/lib
cd /tmp && export TMPDIR=/tmp
for v in "" _walk1; do
    OUT=$ROOT/gpurun_out/prof_${TAG}_walk$v
    mkdir -p $OUT
    export SWMI_LIB=$LIB/libswmi$v.so
    rocprofv3 --kernel-trace --stats -d $OUT -o sg --output-format csv -- python3 $ROOT/bench.py --mode semiglobal --no-cpu-baseline > $OUT/sg.log 2>&1 || echo "trace failed"
    f=$(find $OUT -name "*kernel_stats.csv" | head -1)
    echo "== libswmi$v.so"
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if "sg_" in r[0]:
        print("%-28s calls %3s  avg %.3f ms  min %.3f" % (r[0].split("(anonymous namespace)::")[1].split("(")[0][:28], r[1], float(r[3]) / 1e6, float(r[5]) / 1e6 if len(r) > 5 else 0))
PY
done
