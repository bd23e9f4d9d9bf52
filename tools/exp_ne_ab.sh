#!/bin/bash
# one-pass kernel against the previous kernel family (single-pass register kernel below NE = 32, split kernels from 32) per NE
out=gpurun_out/ne_ab; mkdir -p $out
for ne in $*; do for op in 0 1; do
  QPSIM_COLL_ONEPASS=$op timeout -k 10 300 python bench.py --workload coupled1024ne$ne --steps 6 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > $out/b.json 2> $out/b.err
  python - $out/b.json "NE $ne onepass=$op" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"{sys.argv[2]:18s} step {d['ms_per_step']:.3f} ms  call {r['avg_launch_us']:.0f} us  fp64 {r['fp64_frac']:.3f}  {r['kernel'][:30]}")
PY
done; done
