cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_x3_gpu.py tests/test_nn_ops_gpu.py -q -x -m gpu -k "x3 or conv2d_fwd_bwd or batched or split" 2>&1 | tail -3
for s in 0 1 0 1; do PIXELPICK_X3_SHARE=$s timeout 300 python tools/train_bench.py 2>&1 | tail -1; done
for net in FPN deeplab_r50; do for s in 0 1; do PIXELPICK_X3_SHARE=$s NET=$net timeout 300 python tools/train_bench.py 2>&1 | tail -1; done; done
