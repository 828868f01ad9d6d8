cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
timeout 1800 python -m pytest tests/test_networks_gpu.py tests/test_fpn_configs_gpu.py tests/test_layerwise_parity_gpu.py tests/test_driver_gpu.py tests/test_checkpoint_format_gpu.py -q -x > $O/t_net.txt 2>&1; tail -3 $O/t_net.txt
timeout 900 python -m pytest tests/test_dist_gpu.py -q -x -k "two_rank_train_step" > $O/t_dist.txt 2>&1; tail -2 $O/t_dist.txt
for v in 1 0 1 0; do PIXELPICK_SPARSE_LOWRES_CE=$v NET=FPN STEPS=20 python tools/train_bench.py 2>&1 | tail -1 | sed "s/^/FPN lowres_tail=$v /" >> $O/ab.txt; done; cat $O/ab.txt
