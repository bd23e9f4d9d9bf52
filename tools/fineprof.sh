# per-kernel durations of a library variant: tools/fineprof.sh LIB SHAPE...   (rocprofv3 kernel trace of tools/exp_shapes.py)
R=$GRAFT_REPO_ROOT; out=gpurun_out/r2fine; lib=$1; shift; mkdir -p $R/$out; cd /tmp; export TMPDIR=/tmp
export QPSIM_HIP_LIBRARY=$R/$lib
for shape in "$@"; do
  tag=$(basename $lib .so)_$shape
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/$tag -o s -- python3 $R/tools/exp_shapes.py $shape > $R/$out/$tag.txt 2> $R/$out/$tag.err
  echo "== $tag"; python3 - $(find $R/$out/$tag -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("   ", r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'])
PY
done
