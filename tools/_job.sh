cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w; mkdir -p $O
timeout 1500 python -m pytest tests/test_acq_wide_gpu.py tests/test_acq_gpu.py tests/test_acq_lowres_gpu.py tests/test_fpn_acq_gpu.py tests/test_nn_ops_gpu.py -q -k "acq or mc_ or wide or lowres or fpn" 2>&1 | grep -v "^E  " > $O/t.txt; tail -12 $O/t.txt
