cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_acq_gpu.py tests/test_acq_lowres_gpu.py -q -x -k "quantised or large_k or select_modes or lowres" > $O/t_acq.txt 2>&1; tail -3 $O/t_acq.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o lk -- python $GRAFT_REPO_ROOT/bench.py --mode acq --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --k 6553 > $GRAFT_REPO_ROOT/$O/largek_line.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, json, glob
cur=sqlite3.connect(glob.glob("gpurun_out/r4e/prof/*.db")[0]).cursor()
rows=cur.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name order by sum(end-start) desc").fetchall()
for r in rows[:8]: print(f"{r[1]:5d} avg {r[2]:9.1f} us min {r[3]:9.1f}  {r[0][:90]}")
d=json.loads(open("gpurun_out/r4e/largek_line.json").readline()); print(d["value"], d["ms_per_step"], d["acquisition"]["from_lowres_logits"]["value"])
PY
