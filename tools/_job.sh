cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $O
for v in 0 -2; do
  rm -rf /tmp/p$v; 
  BNBYTES=$v STEPS=20 rocprofv3 --kernel-trace --stats -d /tmp/p$v -o t --output-format csv -- python tools/train_bench.py > /dev/null 2>&1
  f=$(find /tmp/p$v -name "*kernel_stats.csv" | head -1)
  echo "BNBYTES=$v"; grep "bn_fused" $f | cut -d, -f1-4 | head -8
  cp $f $O/stats_$v.csv
done
