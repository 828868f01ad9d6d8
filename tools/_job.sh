cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
PIXELPICK_WGRAD_LATE=0 STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
PIXELPICK_WGRAD_LATE=0 REPLAY=1 STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
REPLAY=1 STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
done
