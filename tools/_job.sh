cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_networks_gpu.py -x -q -k "weight_planes or replay" 2>&1 | tail -15
for m in 0 1 0 1; do PIXELPICK_X3_WEIGHT_PREFETCH=$m STEPS=60 python tools/train_bench.py 2>&1 | tail -1; done
for m in 0 1; do REPLAY=1 PIXELPICK_X3_WEIGHT_PREFETCH=$m STEPS=60 python tools/train_bench.py 2>&1 | tail -1; done
