cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
for v in 0 1 auto 0 auto; do PIXELPICK_FUSE_DW_BN=$v STEPS=30 python tools/train_bench.py 2>&1 | tail -1 | sed "s/^/fuse_dw_bn=$v /" >> $O/ab.txt; done; cat $O/ab.txt
