cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
for r in 0 16 32 64 0 32; do PIXELPICK_COMM_CU_RESERVE=$r STEPS=20 python tools/train_bench.py 2>&1 | tail -1 | sed "s/^/reserve=$r /" >> $O/reserve.txt; done; cat $O/reserve.txt
