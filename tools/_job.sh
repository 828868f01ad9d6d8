cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
timeout 1200 python -m pytest tests/test_networks_gpu.py tests/test_fpn_configs_gpu.py tests/test_layerwise_parity_gpu.py -q -x > $O/t_net.txt 2>&1; tail -3 $O/t_net.txt
for net in FPN deeplab_r50 deeplab; do
  for thr in 0 268435456; do
    python - <<PY >> $O/ab.txt
import os
os.environ["NET"]="$net"
from pixelpick_amd import _lib
_lib.lib().pp_debug_set_conv_thresholds($thr)
import subprocess,sys
PY
    THR=$thr NET=$net STEPS=20 python -c "
import os,sys
sys.argv=['x']
from pixelpick_amd import _lib
_lib.lib().pp_debug_set_conv_thresholds(int(os.environ['THR']))
sys.path.insert(0,'tools')
import train_bench; train_bench.main()" 2>&1 | tail -1 | sed "s/^/$net thr=$thr /" >> $O/ab.txt
  done
done
cat $O/ab.txt
