cd $GRAFT_REPO_ROOT
O=gpurun_out/r4t; mkdir -p $O
( time python bench.py > $O/bench_line.json 2> $O/bench.err ) 2> $O/time.txt; tail -3 $O/time.txt
python -c "
import json; d=json.loads(open('$O/bench_line.json').readline()); t=d['train']; print(d['value'], d['ms_per_step'], t['host_enqueue_ms_per_step'], t['native_replay']); print([ (o['config'][:12], o['value']) for o in d['other_configs']])
print(d['other_configs'][1]['roofline'].get('frac_incl_map_write'))"
timeout 900 python -m pytest tests/test_dist_gpu.py -q -x -k "bench" > $O/t_bench.txt 2>&1; tail -2 $O/t_bench.txt
