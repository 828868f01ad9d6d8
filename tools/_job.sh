cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_acq_gpu.py -x -q -k "full_size_picks" > $O/t_guard.txt 2>&1; tail -5 $O/t_guard.txt
(time python bench.py) > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench_line.json; tail -5 $O/bench.err
