#!/bin/bash
# Timing-only ablation builds of the rectangle ADI kernels: tools/bin/libqpsim_abl<N>.so with -DQP_ABL=<N>
# (bit 0 no ghost loads, 1 constants instead of table loads, 2 no transposes, 3 no dot products / interface stores,
#  4 no Thomas solve, 5 no explicit operator).  Results are wrong by construction; run tools/exp_shapes.py against them:
#    QPSIM_HIP_LIBRARY=tools/bin/libqpsim_abl1.so python tools/exp_shapes.py 4096x4096
set -e
cd "$(dirname "$0")/.."
C=quasiparticle-physics-simulation_amd/csrc
mkdir -p tools/bin
# an argument N:W additionally forces W waves per SIMD (-DQP_FORCE_WAVES=W); the library is then libqpsim_ablN_wW.so
for spec in "$@"; do
  n=${spec%%:*}; w=""; [[ $spec == *:* ]] && w=${spec##*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -mllvm -pragma-unroll-threshold=1000000 \
      -DQP_ABL=$n ${w:+-DQP_FORCE_WAVES=$w} -c $C/qp_adi_rect.hip -o tools/bin/rect_abl$n${w:+_w$w}.o &
done
wait
for spec in "$@"; do
  n=${spec%%:*}; w=""; [[ $spec == *:* ]] && w=${spec##*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libqpsim_abl$n${w:+_w$w}.so \
      tools/bin/rect_abl$n${w:+_w$w}.o $(ls $C/*.o | grep -v qp_adi_rect.o)
done
