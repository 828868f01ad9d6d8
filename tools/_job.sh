# scratch job for `gpurun -- 'bash tools/_job.sh'`
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
timeout 2700 python -m pytest tests/ -q -m gpu -x > $O/tall.txt 2>&1; tail -3 $O/tall.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python tools/dw_bench.py > $O/dw_layers.txt 2> $O/dw_layers.err; tail -3 $O/dw_layers.txt; tail -3 $O/dw_layers.err
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench_line.json; echo; tail -3 $O/bench.err
