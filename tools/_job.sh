cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
for i in 1 2 3; do for w in 1 0; do echo "== deeplab GEMMPW=$w" >> $O/train_ab.txt; NET=deeplab GEMMPW=$w STEPS=30 timeout 600 python tools/train_bench.py >> $O/train_ab.txt 2>&1; done; done; grep -E "==|img" $O/train_ab.txt | tail -20
python bench.py --mode train --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['train']['host_enqueue_ms_per_step'])"
