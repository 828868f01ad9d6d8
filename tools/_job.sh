# scratch job for `gpurun -- 'bash tools/_job.sh'`: generic quantised select - tests + timing against the radix select
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g1; mkdir -p $O
timeout 600 python -m pytest tests/test_acq_gpu.py -x -q > $O/tacq.txt 2>&1; tail -3 $O/tacq.txt
cat > /tmp/t.py <<'P'
import os, sys
os.environ["PIXELPICK_KNOBS_BUILD"] = "1"
import torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pixelpick_amd import _lib, acquisition as acq
L = _lib.lib()
torch.manual_seed(0)
for B, N, k in ((256, 131072, 6553), (32, 131072, 6553), (1, 131072, 6553)):
    m = torch.rand(B, N, device="cuda")
    for mode, name in ((1 << 22, "radix select (4 histogram passes + sort)"), (0, "min/max + quantised select")):
        L.pp_debug_set_reduce_mode(mode)
        for _ in range(3): acq.topk_select(m, k, False)
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); acq.topk_select(m, k, False); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"B={B} N={N} k={k}: {name}: {ts[len(ts)//2]*1e3:.1f} us (median of 20, incl. the output allocation)")
    L.pp_debug_set_reduce_mode(0)
P
timeout 120 python /tmp/t.py 2>&1 | tee $O/generic_select.txt
