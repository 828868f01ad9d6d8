cd $GRAFT_REPO_ROOT
for c in 1024 512 256 128 2048 1024; do echo "wgrad target $c: $(WGRAD_TARGET=$c timeout 300 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-80)"; done
