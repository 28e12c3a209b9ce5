#!/bin/bash
# Run ON THE GPU BOX: occupancy / LDS-size sweep of the packed kernel (BASELINE.json configs[2]).  SWMI_EXTRA_LDS adds dynamic
# LDS to every launch, which lowers the number of workgroups (4 wavefronts each, one per SIMD) a CU can hold.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
PAIRS=${1:-67108864}
for extra in 0 17408 45000 100000; do
  SWMI_EXTRA_LDS=$extra python3 $ROOT/bench.py --pairs $PAIRS --steps 20 --warmup 3 --no-rows --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
lds = 35872 + $extra
print('extra LDS %6d B -> %6d B per workgroup, %d workgroups = wavefronts per SIMD: %8.3f ms per %d pairs, %6.1f M alignments/s, frac %.3f' % ($extra, lds, min(4, 163840 // lds), d['ms_per_step'], $PAIRS, d['value'] / 1e6, r['frac']))"
done
