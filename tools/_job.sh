cd $GRAFT_REPO_ROOT
cp tools/probe/libpp_timing.so pixelpick_amd/libpixelpick_hip_knobs.so
cat > /tmp/t.py <<'P'
import os, sys
os.environ["PIXELPICK_KNOBS_BUILD"] = "1"
import torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pixelpick_amd import _lib
B, C, H, W = 256, 19, 256, 512
k = H * W * 5 // 100
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda") * 3
L = _lib.lib()
idx = torch.empty((B, k), dtype=torch.int32, device="cuda"); val = torch.empty((B, k), device="cuda")
ws = torch.empty(int(L.pp_acq_workspace_bytes(B, C, H, W, k)), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for i in range(4):
    _lib.check(L.pp_acq_score_topk(x.data_ptr(), B, C, H, W, *x.stride(), None, 0, k, idx.data_ptr(), val.data_ptr(), None, ws.data_ptr(), ws.numel(), st), "op")
    torch.cuda.synchronize()
P
timeout 120 python /tmp/t.py 2>&1 | tail -4
