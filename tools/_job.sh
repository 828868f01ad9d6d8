cd $GRAFT_REPO_ROOT
O=gpurun_out/r4v; mkdir -p $O
timeout 1500 python -m pytest tests/test_driver_gpu.py tests/test_dist_gpu.py -q -x -k "not bench and not occupier" > $O/t_drv.txt 2>&1; tail -3 $O/t_drv.txt
N=256 python tools/driver_bench.py 2>&1 | grep -v amdgpu | tee $O/driver_default.txt
