cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e18; mkdir -p $O
timeout 300 python -m pytest tests/test_acq_gpu.py -x -q -k "list_select" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for m in 0 $((32<<12)); do
RMODE=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$m -o t -- python $GRAFT_REPO_ROOT/tools/topk5_bench.py > $GRAFT_REPO_ROOT/$O/out$m.txt 2>&1
cp $(find $GRAFT_REPO_ROOT/$O/prof$m -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/ks_$m.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof$m
echo "RMODE=$m"; grep "^k=" $GRAFT_REPO_ROOT/$O/out$m.txt; grep "pp::" $GRAFT_REPO_ROOT/$O/ks_$m.csv | head -3 | cut -c1-60,100-260 
done
cd $GRAFT_REPO_ROOT; RMODE=2048 timeout 120 python tools/topk5_bench.py 2>&1 | tail -2 | head -1
