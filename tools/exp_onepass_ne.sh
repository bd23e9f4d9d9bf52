#!/bin/bash
# One-pass collision kernel against the split kernels at every instantiated NE >= 32 (1024^2, dynamic phonons), + parity tests.
out=gpurun_out/onepass_ne; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.log
for ne in 32 40 50; do for op in 0 1; do
  QPSIM_COLL_ONEPASS=$op timeout -k 10 300 python bench.py --workload coupled1024ne$ne --steps 10 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > $out/bench_ne${ne}_op$op.json 2> $out/bench_ne${ne}_op$op.err
  python - $out/bench_ne${ne}_op$op.json "ne$ne onepass=$op" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:18s} ms/step {d['ms_per_step']:.3f}  collision call {r['avg_launch_us']:.0f} us  fp64 frac {r['fp64_frac']:.3f}  {r['kernel'][:40]}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
