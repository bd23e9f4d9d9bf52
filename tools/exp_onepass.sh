#!/bin/bash
# One-pass NE = 50 collision kernel: A/B timing against the split kernels and over target-block sizes; $1 = "test" also runs
# the parity tests, "pmc" the counter passes of the default variant.
out=gpurun_out/onepass; mkdir -p $out
if [ "$1" = test ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.log
fi
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --workload coupled1024ne50 --steps 10 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - $out/bench_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:10s} ms/step {d['ms_per_step']:.3f}  collision call {r['avg_launch_us']:.0f} us  fp64 frac {r['fp64_frac']:.3f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run split QPSIM_COLL_ONEPASS=0
run default QPSIM_COLL_ONEPASS=1

if [ "$1" = pmc ]; then
  bash tools/pmc_passes.sh $out/pmc "--workload coupled1024ne50 --steps 4 --warmup 1 --sustained-seconds 0" \
    "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
    "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" > $out/pmc.log 2>&1
  grep -A1 "onepass" $out/pmc.log | head -8
fi
