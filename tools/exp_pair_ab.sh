#!/bin/bash
# A/B on one box: pair kernel with the anti-diagonal occupations parked in LDS (default library) against the all-resident
# build (tools/bin/libqpsim_pair_resident.so), alternating, c3 and c4.
out=gpurun_out/pair_ab; mkdir -p $out
for rep in 1 2; do for wl in c3 c4; do for lib in default resident; do
  if [ $lib = resident ]; then export QPSIM_HIP_LIBRARY=$PWD/tools/bin/libqpsim_pair_resident.so; else unset QPSIM_HIP_LIBRARY; fi
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > $out/b.json 2> $out/b.err
  python - $out/b.json "$wl $lib" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"{sys.argv[2]:14s} ms/step {d['ms_per_step']:.3f}  pair launch {r['avg_launch_us']:.0f} us")
PY
done; done; done
