cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_fpn_configs_gpu.py tests/test_networks_gpu.py -q -m gpu 2>&1 | tail -4
