import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib, engine as E
L = _lib.lib(); dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for name, M, C in [("1/16 18x34x960", 2448, 960), ("1/16 16x32x160", 2048, 160), ("1/16 16x32x64", 2048, 64), ("1/8 34x66x192", 4*34*66, 192), ("head", 4*64*128, 256)]:
    x = torch.randn(M, C, device=dev); y = torch.empty_like(x)
    sc, sh = torch.rand(C, device=dev), torch.rand(C, device=dev)
    sync, ws = E._bn_exchange(dev)
    g, b = torch.rand(C, device=dev), torch.rand(C, device=dev); mean, inv = torch.empty(C, device=dev), torch.empty(C, device=dev)
    def ssa(): _lib.check(L.pp_scale_shift_act(x.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), None, 0, 2, y.data_ptr(), C, st), "ssa")
    def bn(): _lib.check(L.pp_bn_train_fwd_fused(x.data_ptr(), C, M, C, g.data_ptr(), b.data_ptr(), 1e-5, 0.1, None, None, mean.data_ptr(), inv.data_ptr(), None, 0, 2, 0.0, 0, None, y.data_ptr(), C, ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), st), "bn")
    res = []
    for fn in (ssa, bn):
        fn(); ts = []
        # warm (data in L2/MALL as in the real step: the conv just wrote x) and back-to-back pairs to expose the marginal cost
        for _ in range(20):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); 
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 100)
        ts.sort(); res.append(ts[len(ts)//2])
    print(f"{name:20s} scale_shift_act {res[0]:6.1f} us   bn_fused_fwd {res[1]:6.1f} us   (per launch, 10 back to back, warm)")
