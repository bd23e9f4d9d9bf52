#!/bin/bash
# A/B of two builds of the library on the same box, alternating: tools/ab_libs.sh LIB_A LIB_B REPS SHAPE...
A=$1; B=$2; R=$3; shift 3
for r in $(seq $R); do
  for L in $A $B; do echo "== $L"; QPSIM_HIP_LIBRARY=$L python tools/exp_shapes.py "$@"; done
done
