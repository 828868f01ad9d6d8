cd $GRAFT_REPO_ROOT
for c in 0 192 128 96 64; do echo "side CUs $c: $(PIXELPICK_SIDE_CUS=$c timeout 300 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-80)"; done
for c in 128 64; do echo "side CUs $c stride 2: $(PIXELPICK_SIDE_CUS=$c PIXELPICK_SIDE_CU_STRIDE=2 timeout 300 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-80)"; done
echo "side CUs 0: $(timeout 300 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-80)"
