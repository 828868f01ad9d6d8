import sys, torch
sys.path.insert(0, "/root/repo")
from pixelpick_amd import engine as E
import torch.nn.functional as F
torch.manual_seed(0)
def run(M, C, act, fused, x, dy, gamma, beta):
    E._BN_FUSED = fused
    tape = E.Tape()
    xv = E.Var(x.view(1, 1, M, C).clone()); xv.needs_grad = True
    g = gamma.clone().requires_grad_(True); b = beta.clone().requires_grad_(True)
    yv = E.batch_norm_act(tape, xv, g, b, None, None, True, act)
    tape.backward(yv, dy.view(1, 1, M, C))
    return yv.t.view(M, C), xv.grad.view(M, C), tape.param_grads[id(g)], tape.param_grads[id(b)]
for (M, C) in [(768, 192), (2048, 960), (32768, 48), (131072, 32), (3072, 96)]:
    x = (torch.randn(M, C, device="cuda") * 1.5 + 0.3)
    dy = torch.randn(M, C, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    xd = x.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    yd = F.relu6(F.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5))
    yd.backward(dy.double())
    for fused in (True, False):
        y, dx, dg, db = run(M, C, 2, fused, x, dy, gamma, beta)
        # exclude elements whose relu6 mask could differ
        print(M, C, "fused" if fused else "3launch",
              "y %.2e" % (y.double() - yd).abs().max().item(),
              "dx %.2e" % (dx.double() - xd.grad).abs().max().item(),
              "dg %.2e" % ((dg.double() - gd.grad).abs().max() / gd.grad.abs().max()).item(),
              "db %.2e" % ((db.double() - bd.grad).abs().max() / bd.grad.abs().max()).item())
