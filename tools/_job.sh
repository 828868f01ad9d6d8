cd $GRAFT_REPO_ROOT
for m in 0 1 0 1; do PIXELPICK_BATCH_REDUCE=$m STEPS=60 python tools/train_bench.py 2>&1 | tail -1; done
