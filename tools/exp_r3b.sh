#!/bin/bash
# (1) c3 / c4 with the phonon-prefetch depth of the single-pass collision kernel varied (tools/coll_pf_variants.sh);
# (2) where the runtime copy / fill operations of the default-scheme bench sit in time (plan creation or steady state).
out=gpurun_out/r3b; mkdir -p $out
run() { tag=$1; wl=$2; shift 2; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > $out/bench_${wl}_$tag.json 2> $out/bench_${wl}_$tag.err
  python - $out/bench_${wl}_$tag.json $wl-$tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:14s} ms/step {d['ms_per_step']:.3f}  dominant launch {r['avg_launch_us']:.0f} us  frac {r['frac']:.3f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for wl in c3 c4; do
  run pf1 $wl QPSIM_DUMMY=1
  for n in 2 3 4; do [ -f tools/bin/libqpsim_pf$n.so ] && run pf$n $wl QPSIM_HIP_LIBRARY=$PWD/tools/bin/libqpsim_pf$n.so; done
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$out/cntrace -o t -- python3 $R/bench.py --workload cn4096 --steps 20 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > $R/$out/cntrace.json 2> $R/$out/cntrace.err
cd $R; python - $(find $out/cntrace -name "*kernel_trace.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
t0=min(int(r['Start_Timestamp']) for r in rows)
ev=[(int(r['Start_Timestamp'])-t0, r['Kernel_Name'][:40]) for r in rows]
ev.sort()
last=ev[-1][0]
import collections
for name in ('__amd_rocclr_copyBuffer','__amd_rocclr_fillBufferAligned'):
    ts=[t for t,n in ev if n.startswith(name)]
    print(name, len(ts), 'calls; in the last half of the run:', sum(t>last/2 for t in ts), '; in the last quarter:', sum(t>0.75*last for t in ts))
pr=[t for t,n in ev if 'fine_x_kernel' in n]
print('fine_x launches', len(pr), 'span ms', (pr[-1]-pr[0])/1e6)
PY
