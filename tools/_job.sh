cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_networks_gpu.py -x -q -k "dense_labels" 2>&1 | tail -5
