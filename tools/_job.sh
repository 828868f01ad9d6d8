cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_ops_gpu.py -q -m gpu -k "stride2_backward" 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" | head -30
rm -rf /tmp/pt; STEPS=10 rocprofv3 --kernel-trace --stats -d /tmp/pt -o t --output-format csv -- python tools/train_bench.py > /dev/null 2>&1
grep "dwconv_s2_bwd\|dwconv_bwd_data" $(find /tmp/pt -name "*kernel_stats.csv" | head -1) | cut -d, -f1-7
for i in 1 2; do
REPLAY=1 STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1
done
