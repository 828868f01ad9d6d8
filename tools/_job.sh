cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
for b in 4 8 16 32; do B=$b STEPS=10 python tools/train_bench.py 2>&1 | tail -1 | sed "s/^/B=$b /" >> $O/batch.txt; done; cat $O/batch.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs > $GRAFT_REPO_ROOT/$O/bench_line_headline.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_headline_kernel_stats.csv; grep "acq_kernel<19" $O/bench_headline_kernel_stats.csv | cut -c1-160
python -c "
import json; d=json.loads(open('$O/bench_line_headline.json').readline()); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
