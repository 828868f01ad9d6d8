cd $GRAFT_REPO_ROOT
for occ in 0 512 256 0 512; do python bench.py --mode acq --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --tune-occ $occ 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tune $occ', d['acquisition']['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'])"; done
python bench.py --mode acq --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for c in d.get('other_configs',[]):
    if c['leg']=='acquisition': print(c['config'][:72], c['value'], c['ms_per_step'], c['roofline']['frac'], c['roofline']['read_only_yardstick'])
"
