cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t; mkdir -p $O
timeout 900 python -m pytest tests/test_acq_gpu.py tests/test_acq_wide_gpu.py -x -q 2>&1 | tail -3
python - <<'PY'
import os,sys,subprocess
PY
for m in 0 1024 0 1024; do RMODE=$m python tools/topk5_bench.py 2>&1 | grep "op "; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -o x -- python $GRAFT_REPO_ROOT/tools/topk5_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r5t/prof2/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=30: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'])
PY
