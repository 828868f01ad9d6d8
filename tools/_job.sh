cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nn_ops_gpu.py tests/test_blocks_gpu.py tests/test_networks_gpu.py -q -x -m gpu 2>&1 | tail -3
for v in 0 $((1<<23)) 0 $((1<<23)); do echo "conv variant $v"; CONVVAR=$v timeout 300 python tools/train_bench.py 2>&1 | tail -1; done
for v in 0 $((1<<23)); do NET=FPN CONVVAR=$v timeout 300 python tools/train_bench.py 2>&1 | tail -1;  NET=deeplab_r50 CONVVAR=$v timeout 300 python tools/train_bench.py 2>&1 | tail -1; done
