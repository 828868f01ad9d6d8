cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests/test_dist_gpu.py -x -q -k "bench" > $O/t3.txt 2>&1
tail -40 $O/t3.txt
python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 5 --replay on > $O/bench_replay.json 2>$O/bench_replay.err
python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 5 --replay off > $O/bench_eager.json 2>>$O/bench_replay.err
python -c "
import json
for f in ('bench_replay','bench_eager'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['train']['host_enqueue_ms_per_step'], d['train']['launch'])
"
