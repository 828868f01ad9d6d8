cd $GRAFT_REPO_ROOT
O=gpurun_out/r5fin; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS=12 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) --list > $O/train_step_timeline.txt 2>&1; head -12 $O/train_step_timeline.txt
