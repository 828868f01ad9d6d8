cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e19; mkdir -p $O
timeout 600 python -m pytest tests/test_acq_gpu.py -x -q > $O/tacq.txt 2>&1; tail -3 $O/tacq.txt
for m in 2048 0; do echo "RMODE=$m"; RMODE=$m timeout 120 python tools/topk5_bench.py 2>&1 | tail -2 | head -1; done | tee $O/topk5.txt
echo "every image falls back (aim k/16):"; RMODE=$((1<<12)) timeout 120 python tools/topk5_bench.py 2>&1 | tail -2 | head -1
cd /tmp && export TMPDIR=/tmp
RMODE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o t -- python $GRAFT_REPO_ROOT/tools/topk5_bench.py > /dev/null 2>&1
cp $(find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/ks.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof
grep "pp::" $GRAFT_REPO_ROOT/$O/ks.csv | head -5 | cut -c1-60,100-300
