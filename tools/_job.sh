cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
python tools/plan_overhead.py 2>&1 | grep -v amdgpu.ids | tee $O/plan_overhead.txt
