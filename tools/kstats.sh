# per-kernel durations of one bench workload: tools/kstats.sh WORKLOAD [bench args]   (rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; out=gpurun_out/kstats; w=$1; shift; mkdir -p $R/$out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/$w -o s -- python3 $R/bench.py --workload $w --no-cpu-baseline "$@" > $R/$out/$w.json 2> $R/$out/$w.err
python3 - $(find $R/$out/$w -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("   ", r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
