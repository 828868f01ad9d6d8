cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_nn_ops_gpu.py -q -x -m gpu -k "batched or dwconv" 2>&1 | grep -E "differ|passed|failed" | cut -c1-600
for f in 0 4 16 64 768; do if [ $f = 0 ]; then export PIXELPICK_BATCH_REDUCE=0; else export PIXELPICK_BATCH_REDUCE=1 PIXELPICK_REDUCE_FLUSH_MB=$f; fi; echo "flush MB $f"; for i in 1 2; do timeout 300 python tools/train_bench.py 2>&1 | tail -1; done; done
export PIXELPICK_BATCH_REDUCE=1 PIXELPICK_REDUCE_FLUSH_MB=768
cd /tmp; export TMPDIR=/tmp
STEPS=6 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py > $GRAFT_REPO_ROOT/$O/trace_b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) --by-queue --list > $O/timeline_batch.txt 2>&1; head -4 $O/timeline_batch.txt
