cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3u; mkdir -p $O
HOSTPROF=20 STEPS=20 python tools/train_bench.py > $O/hostprof.txt 2>&1
