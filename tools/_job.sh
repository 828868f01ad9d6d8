cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; mkdir -p $O
timeout 900 python -m pytest tests/test_nn_ops_gpu.py -x -q -k "batchnorm or bn_" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
STEPS=12 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py > $GRAFT_REPO_ROOT/$O/train_bench.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) 10 > $O/timeline.txt 2>&1; head -40 $O/timeline.txt; tail -3 $O/train_bench.txt
