cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o; mkdir -p $O
timeout 900 python -m pytest tests/test_wgrad_fold_gpu.py -q 2>&1 | tail -12
for w in 0 33555456; do for i in 1 2; do echo "== deeplab WGRAD_TARGET=$w" ; NET=deeplab WGRAD_TARGET=$w STEPS=30 timeout 600 python tools/train_bench.py 2>&1 | tail -1; done; done
for w in 0 33555456; do echo "== FPN WGRAD_TARGET=$w" ; NET=FPN WGRAD_TARGET=$w STEPS=20 timeout 600 python tools/train_bench.py 2>&1 | tail -1; done
