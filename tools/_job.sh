cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_driver_gpu.py -q -x -m gpu 2>&1 | tail -6
