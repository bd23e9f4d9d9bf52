#!/bin/bash
# same-box A/B of gap-class one-pass kernel variants (tools/bin/libqpsim_gap*.so) at NE = 50, 4 classes
for rep in 1 2; do for lib in default $(ls tools/bin/libqpsim_gap*.so); do
  if [ $lib = default ]; then unset QPSIM_HIP_LIBRARY; else export QPSIM_HIP_LIBRARY=$PWD/$lib; fi
  python bench.py --workload coupled1024ne50gap4 --steps 6 --warmup 2 --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['roofline']['avg_launch_us']), 'us per call')"
done; done
