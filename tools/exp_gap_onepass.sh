#!/bin/bash
# A/B of the gap-class collision call at NE = 50 (4 classes): one-pass kernel vs the three split kernels, + parity tests.
set -e
for rep in 1 2; do for mode in split onepass; do
  if [ $mode = split ]; then export QPSIM_COLL_ONEPASS=0; else unset QPSIM_COLL_ONEPASS; fi
  python bench.py --workload coupled1024ne50gap4 --steps 6 --warmup 2 --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['roofline']['avg_launch_us']), 'us per call', d['roofline'].get('kernel'))"
done; done
unset QPSIM_COLL_ONEPASS
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gap_classes" -x 2>&1 | tail -3
