cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests/ -q -m gpu > $O/tall.txt 2>&1
tail -5 $O/tall.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma']['frac'])
for r in d['roofline_mfma_1x1']['shapes']: print(r['shape'][:28], r['us'], r['library_sgemm_us'], r['frac_of_mfma_peak'], r['vs_library'])
"
