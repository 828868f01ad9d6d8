cd $GRAFT_REPO_ROOT
O=gpurun_out/r5fin4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/$O/prof_head
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_head -o b -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs > $GRAFT_REPO_ROOT/$O/headline_line_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_head -name "*kernel_stats.csv" | head -1) $O/bench_headline_kernel_stats.csv
grep "acq_kernel<19" $O/bench_headline_kernel_stats.csv | cut -c1-200
python -c "
import json; d=json.load(open('$O/headline_line_under_rocprof.json')); print(d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['value'])"
