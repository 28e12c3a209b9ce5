#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.  Outputs under gpurun_out/prof_<tag>/.
# PMC counters are collected in their own runs (never together with --sys-trace etc.), one TCC counter per pass.
set -u
TAG=${1:-r01}
EXTRA=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rows $EXTRA"
# the kernel-trace pass runs the DEFAULT bench command (50 steps, 5 warmup) so that its per-kernel average is comparable
# with the roofline.kernel_ms the same command prints
BENCH_DEFAULT="python3 $ROOT/bench.py --no-cpu-baseline --no-rows $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- $BENCH_DEFAULT > $OUT/trace.log 2>&1 || echo "trace failed"
# --stats averages every launch, the cold ones after start-up included: the steady-state figures (median, mean without the
# first 10 launches) are what compares with the bench's kernel_ms
KT=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python3 $ROOT/tools/summarize_kernel_trace.py $KT 10 > $ROOT/gpurun_out/${TAG}_kernel_steady.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o $TAG --output-format csv -- $BENCH > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM -d $OUT/pmc_sq2 -o $TAG --output-format csv -- $BENCH > $OUT/pmc_sq2.log 2>&1 || echo "pmc_sq2 failed"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o $TAG --output-format csv -- $BENCH > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o $TAG --output-format csv -- $BENCH > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
find $OUT -name "*.csv" | head -30
# derived metrics (ROCm 7.2 has no gfx950 section in derived_counters.xml: these fall back to the gfx94x formulas)
rocprofv3 --pmc VALUBusy VALUUtilization -d $OUT/pmc_derived -o $TAG --output-format csv -- $BENCH > $OUT/pmc_derived.log 2>&1 || echo "pmc_derived failed"
rocprofv3 --pmc MemUnitBusy OccupancyPercent -d $OUT/pmc_derived2 -o $TAG --output-format csv -- $BENCH > $OUT/pmc_derived2.log 2>&1 || echo "pmc_derived2 failed"
