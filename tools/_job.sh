cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
timeout 1500 python -m pytest tests/test_nn_ops_gpu.py tests/test_blocks_gpu.py tests/test_networks_gpu.py -q -x 2>&1 | tail -4
timeout 600 python tools/dw_bench.py > $O/dw_layers.txt 2> $O/dw.err; grep -E "x960  1 0 2|130x258|128x256" $O/dw_layers.txt | cut -c1-140
for i in 1 2 3; do NET=deeplab STEPS=30 timeout 600 python tools/train_bench.py 2>&1 | tail -1; done
