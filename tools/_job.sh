cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; mkdir -p $O
timeout 1500 python -m pytest tests/test_acq_gpu.py tests/test_acq_lowres_gpu.py -q -x > $O/t_acq.txt 2>&1; tail -2 $O/t_acq.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/measure_acq_traffic.py > $O/traffic.txt 2>&1; tail -c 300 $O/traffic.txt; echo
python bench.py > $O/bench_line.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_line.json').readline()); print(d['value'], d['acquisition']['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel_ms_avg'])
for o in d['other_configs']: print(o['config'][:40], o['value'], o.get('roofline',{}).get('frac'))"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_headline_kernel_stats.csv; grep "acq_kernel<19" $O/bench_headline_kernel_stats.csv | cut -c1-170
