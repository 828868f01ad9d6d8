cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 2700 python -m pytest tests/ -q -m gpu > $O/tall.txt 2>&1; tail -3 $O/tall.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_line.json 2> $O/bench.err; python - <<PY
import json; d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["train"]["replay"], d["train"]["host_enqueue_ms_per_step"])
PY
