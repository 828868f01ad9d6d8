# scratch job for `gpurun -- 'bash tools/_job.sh'`
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_fpn_acq_gpu.py tests/test_acq_lowres_gpu.py -x -q -s > $O/t_fpn.txt 2>&1; tail -15 $O/t_fpn.txt
timeout 900 python tools/query_bench.py --configs > $O/qb_configs.txt 2>&1; cat $O/qb_configs.txt | tail -8
