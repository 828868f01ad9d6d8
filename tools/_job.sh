cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_nn_ops_gpu.py -q -m gpu -k "narrow_layers or stem_weight" 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -30
for i in 1 2 3; do
ROWS=1 STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
ROWS=2 STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
done
