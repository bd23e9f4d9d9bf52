#!/bin/bash
# default-scheme (unsplit CN) step time on rectangles + the tests that cover it
for w in cn4096 cn1024 cn2048; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:12], round(d['ms_per_step'],4), 'ms/step, J =', d['roofline']['iterations_per_step'])"
done
timeout 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fine_tiles.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
