#!/bin/bash
# Evidence run of one round on the GPU box: kernel-trace stats and PMC traffic of the headline and of the other measured
# workloads.  tools/profile_round.sh OUTDIR [NAME...]  (OUTDIR relative to the repo root; NAMEs restrict the run to those
# workloads).  Summaries are then copied to profiles/ by hand.
out=$1; shift; only=" $* "
want() { [[ "$only" == "  " || "$only" == *" $1 "* ]]; }
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/$out; cd /tmp; export TMPDIR=/tmp
stats() {   # name, bench args
  want $1 || return 0
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/$1_stats -o s -- python3 $R/bench.py $2 --no-cpu-baseline --sustained-seconds 0 > $R/$out/$1_bench_under_rocprof.json 2> $R/$out/$1_stats.err
  cp $(find $R/$out/$1_stats -name "*kernel_stats.csv" | head -1) $R/$out/$1_kernel_stats.csv
}
traffic() { # name, bench args
  want $1 || return 0
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$out/$1_fetch -o f -- python3 $R/bench.py $2 --no-cpu-baseline --sustained-seconds 0 > /dev/null 2> $R/$out/$1_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$out/$1_write -o w -- python3 $R/bench.py $2 --no-cpu-baseline --sustained-seconds 0 > /dev/null 2> $R/$out/$1_write.err
  (cd $R && python tools/pmc_summary.py $(find $out/$1_fetch -name "*counter_collection.csv") $(find $out/$1_write -name "*counter_collection.csv") $out/$1_pmc.json > $out/$1_pmc.txt 2>&1)
}
stats adi4096 "--steps 50 --warmup 10"
traffic adi4096 "--steps 20 --warmup 3"
stats ring4096 "--workload ring4096 --steps 30 --warmup 5"
traffic ring4096 "--workload ring4096 --steps 20 --warmup 3"
stats c3 "--workload c3 --steps 10 --warmup 2"
traffic c3 "--workload c3 --steps 5 --warmup 1"
stats c2 "--workload c2 --steps 30 --warmup 5"
stats adi8192 "--workload adi8192 --steps 20 --warmup 3"
stats adi2048 "--workload adi2048 --steps 200 --warmup 20"
stats adi1024 "--workload adi1024 --steps 400 --warmup 40"
traffic adi1024 "--workload adi1024 --steps 100 --warmup 10"
traffic adi2048 "--workload adi2048 --steps 50 --warmup 5"
stats coupled1024ne50 "--workload coupled1024ne50 --steps 6 --warmup 2"
traffic coupled1024ne50 "--workload coupled1024ne50 --steps 4 --warmup 1"
stats coupled1024ne50gap4 "--workload coupled1024ne50gap4 --steps 6 --warmup 2"
stats c4 "--workload c4 --steps 20 --warmup 3"
stats cn4096 "--workload cn4096 --steps 20 --warmup 3"
cd $R && cat $out/*_pmc.txt | cut -c1-170
