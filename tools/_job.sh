cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, os, sys
sys.path.insert(0, os.getcwd())
from pixelpick_amd import acquisition as acq, _lib
L=_lib.lib()
torch.manual_seed(0)
B,C,h,w,H,W=256,19,64,128,256,512
low=torch.randn(B,h,w,C,device='cuda')*3
k=H*W*5//100
for _ in range(40): acq.score_topk_lowres(low,(H,W),None,'entropy',k)
for mode in (0,1024,0,1024,0,1024):
    L.pp_debug_set_reduce_mode(mode)
    for _ in range(5): acq.score_topk_lowres(low,(H,W),None,'entropy',k)
    ts=[]
    for _ in range(30):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); acq.score_topk_lowres(low,(H,W),None,'entropy',k); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); print('lowres top-5% mode',mode,'op us',round(ts[15]*1e3,1), 'min', round(ts[0]*1e3,1))
L.pp_debug_set_reduce_mode(0)
PY
