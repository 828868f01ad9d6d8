cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 1500 python -m pytest tests/test_acq_gpu.py tests/test_acq_lowres_gpu.py tests/test_fpn_acq_gpu.py -x -q > $O/t_acq.txt 2>&1; tail -5 $O/t_acq.txt
timeout 600 python -m pytest tests/test_networks_gpu.py -x -q -k "regrows" > $O/t_regrow.txt 2>&1; tail -3 $O/t_regrow.txt
python bench.py --mode acq --no-cpu-baseline --steps 30 > $O/bench_acq.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r5e/bench_acq.json').read().strip().splitlines()[-1])
a=l['acquisition']; print('acq', a['value'], a['ms_per_step'], 'lowres', a['from_lowres_logits']['ms_per_step'], 'exact', a['exact_formula']['ms_per_step'])
r=l['roofline']; print('kernel', r['kernel_ms_avg'], r['frac'], 'oplevel frac', r['algorithmic_bytes_per_launch']/(a['ms_per_step']*1e-3)/8e12)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --mode acq --no-cpu-baseline --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/acq_kernel_stats.csv; head -8 $O/acq_kernel_stats.csv | cut -c1-200
