cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_nn_ops_gpu.py tests/test_networks_gpu.py tests/test_fpn_configs_gpu.py -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
NET=FPN STEPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1
grep "bias_grad_partial\|stem7" $(find /tmp/fp -name "*kernel_stats.csv" | head -1) | cut -c1-160
cd $GRAFT_REPO_ROOT; for i in 1 2; do NET=FPN STEPS=20 python tools/train_bench.py 2>&1 | tail -1; done
