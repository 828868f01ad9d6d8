cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_x3_gpu.py -q -x -m gpu 2>&1 | tail -3
MODES=1,17,1,17 timeout 300 python tools/x3_bench.py 2>&1 | grep "mode" | grep -v wgrad
