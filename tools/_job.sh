cd $GRAFT_REPO_ROOT
O=gpurun_out/r6x1; mkdir -p $O
timeout 1500 python -m pytest tests/test_conv_x3f_gpu.py tests/test_networks_gpu.py tests/test_driver_gpu.py tests/test_fpn_configs_gpu.py tests/test_release_build_gpu.py -x -q 2>&1 | tail -5
