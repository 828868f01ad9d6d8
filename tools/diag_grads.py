import os, sys, warnings
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import formula_init as fi
from tests.test_networks_gpu import _build
net, tag = sys.argv[1], sys.argv[2]
g = np.load(f"tests/golden/net_{'deeplab' if net == 'deeplab' else 'fpn'}_{tag}.npz")
B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
m = _build(C, net).train()
x = fi.formula_input(B, H, W, key=f"x{tag}").cuda()
y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}").cuda()
pred = m(x)["pred"]
loss = F.cross_entropy(pred, y, ignore_index=ign)
loss.backward()
named = dict(m.named_parameters())
rows = []
for i, name in enumerate(g["grad_names"]):
    got = fi.summarize(named[str(name)].grad); ref = g["grad_summary"][i]; noise = g["grad_noise"][i]
    r = [abs(got[j] - ref[j]) / (1e-3 * max(ref[s], 1e-12) + 4 * noise[j]) for j, s in ((1, 1), (2, 2), (0, 1))]
    rows.append((max(r), str(name), r, [abs(got[j]-ref[j])/max(ref[s],1e-12) for j,s in ((1,1),(2,2),(0,1))]))
rows.sort(reverse=True)
print("params over tolerance:", sum(1 for r in rows if r[0] > 1), "of", len(rows))
for r in rows[:12]:
    print(f"{r[1]:55s} err/tol (abs-sum,max,sum) = " + " ".join(f"{v:.2f}" for v in r[2]) + "   rel dev " + " ".join(f"{v:.1e}" for v in r[3]))
print("median err/tol", np.median([r[0] for r in rows]))
