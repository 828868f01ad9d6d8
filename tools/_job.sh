cd $GRAFT_REPO_ROOT
O=gpurun_out/r4z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/measure_acq_traffic.py > $O/traffic.txt 2>&1; tail -c 200 $O/traffic.txt; echo
python bench.py > $O/bench_line.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_line.json').readline()); print(d['value'], d['acquisition']['value'], d['roofline']['frac'], d['roofline']['traffic'], d['acquisition']['from_lowres_logits']['value'])
for o in d['other_configs']: print(o['config'][:40], o['value'], o.get('roofline',{}).get('frac'))"
