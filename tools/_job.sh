cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nn_ops_gpu.py -q -x -m gpu -k "conv2d_fwd_bwd" 2>&1 | tail -3
python bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['roofline_mfma_1x1']['shapes']: print(r['shape'][:40], r['us'], r['library_sgemm_us'], r['frac_of_mfma_peak'])
print(d['value'], d['ms_per_step'])"
for v in 0 $((1<<23)); do echo "conv variant $v"; CONVVAR=$v timeout 300 python tools/train_bench.py 2>&1 | tail -1; NET=deeplab_r50 CONVVAR=$v timeout 300 python tools/train_bench.py 2>&1 | tail -1; done
