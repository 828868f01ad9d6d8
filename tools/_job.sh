cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/measure_acq_traffic.py > $O/traffic.txt 2>&1; tail -c 400 $O/traffic.txt; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/bench_line_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; head -5 $O/bench_kernel_stats.csv | cut -c1-200
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 200 $O/bench_line.json
