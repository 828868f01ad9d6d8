cd $GRAFT_REPO_ROOT
for s in 0 1 0 1 0 1; do echo "fuse $s: $(PIXELPICK_CONV_BN_FUSE=$s timeout 120 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-100)"; done
