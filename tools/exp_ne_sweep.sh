#!/bin/bash
# collision call time of the coupled 1024^2 workload over NE (which kernel family serves which size, and how well)
out=gpurun_out/ne_sweep; mkdir -p $out
for ne in 8 12 16 18 20 24 30 32 40 50; do
  timeout -k 10 300 python bench.py --workload coupled1024ne$ne --steps 6 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > $out/b_$ne.json 2> $out/b_$ne.err
  QPSIM_COLL_PAIR=0 timeout -k 10 300 python bench.py --workload coupled1024ne$ne --steps 6 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > $out/s_$ne.json 2> $out/s_$ne.err
  python - $out/b_$ne.json $out/s_$ne.json $ne <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
s=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); q=s["roofline"]
print(f"NE {sys.argv[3]:3s} step {d['ms_per_step']:.3f} ms (unfused {s['ms_per_step']:.3f})  single call {q['avg_launch_us']:.0f} us  hbm {q['hbm_frac']:.2f} fp64 {q['fp64_frac']:.2f}  {q['kernel'][:34]}")
PY
done
