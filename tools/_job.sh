cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_x3_gpu.py -q 2>&1 | tail -2
ONLY=64x128x3 python tools/conv_layer_table.py 2>/dev/null | tail -3
for i in 1 2 3; do python tools/train_bench.py 2>/dev/null | tail -1; done
