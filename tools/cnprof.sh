# per-kernel durations of the default (exact-CN) step: tools/cnprof.sh N   (rocprofv3 kernel trace of tools/exp_cn.py)
R=$GRAFT_REPO_ROOT; out=gpurun_out/cnprof; mkdir -p $R/$out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/cn$1 -o s -- python3 $R/tools/exp_cn.py $1 > $R/$out/cn$1.txt 2> $R/$out/cn$1.err
cp $(find $R/$out/cn$1 -name "*kernel_stats.csv" | head -1) $R/$out/cn$1_kernel_stats.csv
cat $R/$out/cn$1.txt; head -8 $R/$out/cn$1_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
