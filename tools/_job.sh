cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 1200 python -m pytest tests/test_nn_ops_gpu.py -q -x -k "sparse" > $O/t_sparse.txt 2>&1; tail -3 $O/t_sparse.txt
timeout 1500 python -m pytest tests/test_networks_gpu.py tests/test_fpn_configs_gpu.py tests/test_layerwise_parity_gpu.py tests/test_dist_gpu.py -q -x -k "not occupier and not bench" > $O/t_net.txt 2>&1; tail -3 $O/t_net.txt
for net in FPN deeplab_r50 deeplab; do
  for sw in 1 0; do
    PIXELPICK_SPARSE_WGRAD=$sw NET=$net STEPS=20 python tools/train_bench.py 2>&1 | tail -1 | sed "s/^/$net sparse_wgrad=$sw /" >> $O/ab.txt
  done
  PIXELPICK_SPARSE_ROWS=0 NET=$net STEPS=20 python tools/train_bench.py 2>&1 | tail -1 | sed "s/^/$net sparse_rows=0 /" >> $O/ab.txt
done
cat $O/ab.txt
