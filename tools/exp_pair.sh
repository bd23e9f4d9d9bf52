#!/bin/bash
# Double half-step collision pass: parity tests, then A/B of the coupled BASELINE workloads with and without it.
out=gpurun_out/pair; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_distributed.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $out/pytest.log
for wl in c2 c3 c4; do for pair in 0 1; do
  QPSIM_COLL_PAIR=$pair timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > $out/bench_${wl}_pair$pair.json 2> $out/bench_${wl}_pair$pair.err
  python - $out/bench_${wl}_pair$pair.json "$wl pair=$pair" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:12s} ms/step {d['ms_per_step']:.3f}  value {d['value']:.3e}  single collision call {r['avg_launch_us']:.0f} us")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
