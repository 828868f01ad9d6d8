cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
timeout 2700 python -m pytest tests/ -q -m gpu > $O/tall.txt 2>&1; tail -2 $O/tall.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_line.json 2> $O/bench.err
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats -d /tmp/pb -o t --output-format csv -- python bench.py --no-cpu-baseline > $O/bench_prof_line.json 2>/dev/null
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf /tmp/pt; STEPS=10 rocprofv3 --kernel-trace --stats -d /tmp/pt -o t --output-format csv -- python tools/train_bench.py > /dev/null 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python tools/timeline.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) --list > $O/train_timeline_list.txt 2>&1
python bench.py --network FPN --mode train --no-cpu-baseline > $O/fpn_line.json 2>$O/fpn.err
python bench.py --network deeplab_r50 --mode train --no-cpu-baseline > $O/r50_line.json 2>$O/r50.err
for f in bench_line fpn_line r50_line; do python - <<PY
import json; d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); t=d.get("train",{}); print("$f", d["value"], d["ms_per_step"], t.get("replay"), t.get("host_enqueue_ms_per_step"))
PY
done
