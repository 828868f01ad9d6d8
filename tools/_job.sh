# scratch job for `gpurun -- 'bash tools/_job.sh'`: the round-end checks (GPU suite, smoke, bench line + rocprofv3 stats of the same command)
cd $GRAFT_REPO_ROOT
O=gpurun_out/rfin; mkdir -p $O
timeout 2700 python -m pytest tests/ -q -m gpu > $O/tall.txt 2>&1; tail -2 $O/tall.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/measure_acq_traffic.py > $O/traffic.log 2>&1; tail -2 $O/traffic.log | cut -c1-200; cp profiles/acq_traffic.json $O/acq_traffic.json
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/bench_line_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_head -o b -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_step -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find $O/prof_head -name "*kernel_stats.csv" | head -1) $O/bench_headline_kernel_stats.csv
python tools/timeline.py $(find $O/prof_step -name "*kernel_trace.csv" | head -1) --list > $O/train_step_timeline.txt 2>&1; head -4 $O/train_step_timeline.txt
rm -rf $O/prof_bench $O/prof_head $O/prof_step
