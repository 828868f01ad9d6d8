cd $GRAFT_REPO_ROOT
timeout 120 python - <<'PY'
import torch, os, sys
sys.path.insert(0, '.')
from pixelpick_amd import engine as E, _lib
dev = torch.device('cuda:0')
def run(fuse, B=4, H=16, W=32, Cin=64, Cout=384, act=2, res=False):
    E._CONV_BN_FUSE = fuse
    E._WS_BYTES.clear()
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = (torch.randn(1, 1, Cin, Cout, device=dev, generator=g) / Cin ** 0.5).requires_grad_(True)
    gamma = (torch.rand(Cout, device=dev, generator=g) + 0.5).requires_grad_(True); beta = torch.randn(Cout, device=dev, generator=g).requires_grad_(True)
    rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    r = E.Var(torch.randn(B, H, W, Cout, device=dev, generator=g)) if res else None
    tape = E.Tape()
    xv = E.Var(x)
    c = E.conv2d(tape, xv, w, None, 1, 0, 1)
    d = c._pending is not None
    y = E.batch_norm_act(tape, c, gamma, beta, rm, rv, True, act, r)
    dy = torch.randn(y.t.shape, device=dev, generator=g)
    tape.backward(y, dy)
    torch.cuda.synchronize()
    return d, (y.t.clone(), rm.clone(), rv.clone(), xv.grad.clone(), tape.param_grads[id(w)].clone(), tape.param_grads[id(gamma)].clone())
for kw in (dict(Cin=960, Cout=160, act=0, res=True), dict(Cin=384, Cout=64, act=0), dict(Cin=320, Cout=256, act=1), dict(Cin=960, Cout=320, act=0), dict(Cin=576, Cout=96, act=0, res=True), dict(B=3, H=23, W=30, Cin=576, Cout=96, act=0)):
    d1, a = run(True, **kw); d0, b = run(False, **kw)
    print(kw, d1, d0, ['%.1e' % float((u - v).abs().max() / (v.abs().max() + 1e-12)) for u, v in zip(a, b)])
PY
echo "rc $?"
for s in 0 1 0 1; do echo "fuse $s: $(PIXELPICK_CONV_BN_FUSE=$s timeout 120 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-100)"; done
