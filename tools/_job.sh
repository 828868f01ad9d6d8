cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_conv_x3_gpu.py tests/test_nn_ops_gpu.py -q -m gpu -x 2>&1 | tail -3
