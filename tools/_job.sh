cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_acq_gpu.py -q -x -k "block_order or full_size" > $O/t_acq.txt 2>&1; tail -3 $O/t_acq.txt
bash tools/acq_sweep.sh $O
