#!/bin/bash
# BASELINE config 3: occupancy / LDS-size sweep. Unused dynamic LDS per workgroup caps the workgroups resident per CU.
for L in ${LANES_LIST:-4 8}; do
  for X in 0 8192 16384 24576 36864 49152 65536 81920 122880; do
    echo -n "extra_lds=$X "; SWMI_EXTRA_LDS=$X SCHED="$L:0" python tools/gpu_quick2.py 2>&1 | grep parity
  done
done
