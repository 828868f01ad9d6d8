cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 600 python tools/bn_probe.py 2>&1 | grep -v amdgpu.ids > $O/bn_probe.txt
timeout 900 python -m pytest tests/test_nn_ops_gpu.py tests/test_networks_gpu.py -q -x -m gpu 2>&1 | tail -3
for s in 1 2 3; do timeout 300 python tools/train_bench.py 2>&1 | tail -1; done
