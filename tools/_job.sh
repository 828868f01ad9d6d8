cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3z; mkdir -p $O
timeout 1200 python -m pytest tests/test_nn_ops_gpu.py -q -m gpu -k "last_consumer or closed or consumer_conv or stem_weight" 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -30
for i in 1 2 3; do STEPS=200 timeout 600 python tools/train_bench.py 2>&1 | tail -1; done
rm -rf /tmp/pt; STEPS=10 rocprofv3 --kernel-trace -d /tmp/pt -o t --output-format csv -- python tools/train_bench.py > /dev/null 2>&1
python tools/timeline.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) --list > $O/list.txt 2>&1
