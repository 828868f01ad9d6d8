"""Shared by tools/gen_golden_net_trace.py (applied to the imported reference) and tests/test_oracle_trace_golden.py
(applied to oracle/net.py): per-module forward / arriving-gradient summaries of a torch model, keyed by module name."""
import numpy as np
import torch

_KINDS = (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.GroupNorm)
N_SAMPLES = 8


def _summ(t: torch.Tensor) -> np.ndarray:
    """[sum, abs-sum, abs-max, 8 evenly strided samples] in float64."""
    v = t.detach().double().reshape(-1)
    step = max(v.numel() // N_SAMPLES, 1)
    s = v[::step][:N_SAMPLES]
    if s.numel() < N_SAMPLES:
        s = torch.cat([s, torch.zeros(N_SAMPLES - s.numel(), dtype=torch.float64)])
    return np.concatenate([[v.sum().item(), v.abs().sum().item(), v.abs().max().item()], s.numpy()])


def trace_model(model: torch.nn.Module, loss_fn) -> dict:
    """Runs loss_fn() (forward) and backward with hooks on every Conv2d / BatchNorm2d / GroupNorm of `model`.  Summaries
    are taken INSIDE the hooks: the reference's in-place ReLU / `out += residual` modify the hooked tensors afterwards
    (a tensor hook registered before an in-place op receives the gradient of the value before it)."""
    fwd, grad, hooks = {}, {}, []

    def make(name):
        def hook(mod, inp, out):
            assert name not in fwd, f"{name} executed twice"
            fwd[name] = _summ(out)
            out.register_hook(lambda g, n=name: grad.__setitem__(n, _summ(g)))
        return hook

    for name, mod in model.named_modules():
        if isinstance(mod, _KINDS):
            hooks.append(mod.register_forward_hook(make(name)))
    loss = loss_fn()
    loss.backward()
    for h in hooks:
        h.remove()
    names = list(fwd.keys())
    pn = [n for n, _ in model.named_parameters()]
    return {"names": np.array(names), "fwd": np.stack([fwd[n] for n in names]), "grad": np.stack([grad[n] for n in names]),
            "param_names": np.array(pn), "param_grad": np.stack([_summ(p.grad) for _, p in model.named_parameters()]),
            "loss": np.float64(loss.item())}
