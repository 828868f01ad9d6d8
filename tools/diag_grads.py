import os, sys, warnings
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import formula_init as fi
from tests.test_networks_gpu import _build
tag = sys.argv[1] if len(sys.argv) > 1 else "cs64x96"
g = np.load(f"tests/golden/net_deeplab_{tag}.npz")
B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
m = _build(C).train()
x = fi.formula_input(B, H, W, key=f"x{tag}").cuda()
y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}").cuda()
pred = m(x)["pred"]
loss = F.cross_entropy(pred, y, ignore_index=ign)
loss.backward()
named = dict(m.named_parameters())
for i, name in enumerate(g["grad_names"]):
    got = fi.summarize(named[str(name)].grad); ref = g["grad_summary"][i]
    print(f"{str(name):55s} abs-sum rel {abs(got[1]-ref[1])/max(ref[1],1e-12):.2e}  max rel {abs(got[2]-ref[2])/max(ref[2],1e-12):.2e}  sum/abs {abs(got[0]-ref[0])/max(ref[1],1e-12):.2e}")
