cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
timeout 1200 python -m pytest tests/test_nn_ops_gpu.py -q -m gpu -k "narrow_layers" 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -30
for i in 1 2 3; do
CONVVAR=16777216 STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
done
rm -rf /tmp/pt; STEPS=10 rocprofv3 --kernel-trace --stats -d /tmp/pt -o t --output-format csv -- python tools/train_bench.py > /dev/null 2>&1
grep "rows_kernel\|widen" $(find /tmp/pt -name "*kernel_stats.csv" | head -1) | cut -d, -f1-7
