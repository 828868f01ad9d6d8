cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
for i in 1 2; do timeout 300 python tools/train_bench.py 2>&1 | tail -1; NO_WGRAD=1 timeout 300 python tools/train_bench.py 2>&1 | tail -1; done
cd /tmp; export TMPDIR=/tmp
STEPS=6 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) --by-queue --list > $O/timeline.txt 2>&1; head -5 $O/timeline.txt
