cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_x3_gpu.py -x -q -k "row_staged" > $O/t_rows.txt 2>&1; tail -5 $O/t_rows.txt
FWD_ONLY=1 MODES=9,1 timeout 300 python tools/x3_bench.py > $O/x3_bench.txt 2>&1; cat $O/x3_bench.txt
