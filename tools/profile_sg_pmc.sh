#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PMC passes of the semi-global bench mode (row N4), one counter group per run, never
# together with a trace.  Outputs under gpurun_out/prof_sg/; tools/summarize_sg_pmc.py turns them into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_sg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --mode semiglobal --steps 2 --warmup 1 --no-cpu-baseline --sg-plain"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o sg --output-format csv -- $BENCH > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o sg --output-format csv -- $BENCH > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o sg --output-format csv -- $BENCH > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
find $OUT -name "*counter_collection.csv" | head
