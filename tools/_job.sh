cd $GRAFT_REPO_ROOT
timeout 300 python tools/_dbg.py 2>&1 | grep -v amdgpu | tail -8
O=gpurun_out/r3n; mkdir -p $O
timeout 2700 python -m pytest tests/ -q -m gpu > $O/tall.txt 2>&1; tail -4 $O/tall.txt
