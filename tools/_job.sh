cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tr
NET=FPN STEPS=5 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py > $GRAFT_REPO_ROOT/$O/trace_FPN2.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/tr -name "*kernel_stats.csv" | head -1) $O/fpn_kernel_stats.csv
head -30 $O/fpn_kernel_stats.csv | cut -c1-150
