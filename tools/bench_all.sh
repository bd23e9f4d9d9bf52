#!/bin/bash
# Runs the benchmark workloads of DESIGN.md section 6 on one GPU; one JSON line per workload under $1.
out=$1; mkdir -p $out
python bench.py > $out/bench_adi4096.json 2> $out/bench_adi4096.err
for w in adi8192 adi2048 adi1024 adi256x768 cn4096 cn1024 c2 c3 c4 ring4096 ring4096x12 coupled1024ne32 coupled1024ne40 coupled1024ne50 coupled1024ne50gap4 dd8192 dd8192c ddx8192; do
  python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err
done
python bench.py --force-dist --force-subrecords --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_subrecords_1rank.json 2> $out/bench_subrecords_1rank.err
tail -n 3 $out/*.err | grep -v amdgpu.ids | grep -v "^$" | tail -40
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f"{sys.argv[1].split('bench_')[-1][:-5]:22s} value {d['value']:.3e} ms/step {d['ms_per_step']:.4f} roof {r['bound']} {r['frac']:.3f} ({r['avg_launch_us']:.1f} us)  step-frac {d['hbm_frac_of_step']:.3f}")
    for k in ("strong","ensemble"):
        if k in d: print("   ",k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in d[k].items() if a not in ("workload","path")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
