cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats -d /tmp/pb -o t --output-format csv -- python bench.py --no-cpu-baseline > $O/bench_prof_line.json 2>/dev/null
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf /tmp/pt; STEPS=10 rocprofv3 --kernel-trace --stats -d /tmp/pt -o t --output-format csv -- python tools/train_bench.py > /dev/null 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python tools/timeline.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) > $O/train_timeline.txt 2>&1
STEPS=3000 python tools/train_bench.py 2>&1 | tail -1
python - <<PY
import json; d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["train"]["replay"], d["train"]["host_enqueue_ms_per_step"])
PY
