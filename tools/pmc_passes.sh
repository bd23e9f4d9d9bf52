#!/bin/bash
# Separate rocprofv3 --pmc passes (kernel trace only, as the pool requires) of one bench.py workload, summarised per kernel.
#   tools/pmc_passes.sh OUTDIR "BENCH ARGS" "COUNTERS PASS 1" "COUNTERS PASS 2" ...
# run from /tmp with TMPDIR=/tmp (rocprofv3 scratch); python3 is the profiled program itself (no wrappers).
out=$1; args=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/$out; cd /tmp; export TMPDIR=/tmp
n=0
for pass in "$@"; do
  n=$((n+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$out/pass$n -o p -- python3 $R/bench.py $args --no-cpu-baseline > $R/$out/pass$n.log 2>&1
done
cd $R && python tools/pmc_counters.py $out $(find $out -name "*counter_collection.csv" | sort)
