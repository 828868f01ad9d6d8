cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 600 python tools/dw_wgrad_bench.py 2>&1 | grep -v amdgpu.ids > $O/dw_wgrad.txt; cat $O/dw_wgrad.txt
