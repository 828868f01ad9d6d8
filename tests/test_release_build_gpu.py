"""The suite runs on the TEST BUILD (tests/conftest.py: libpixelpick_hip_knobs.so, the product's sources + -DPP_DEBUG_KNOBS, because
the parity tests force kernel forms through the pp_debug_* planner switches).  This file holds the PRODUCT library
(libpixelpick_hip.so: include/pixelpick_hip.h and nothing else) to the same results: one script, run in a process of its own on each
build - acquisition (query.py:190-204,57-61; three strategies, default and reference-order scorer, k = 20 and the top-5 % mode) and
three DeepLabv3+-MNv2 train steps (model.py:101-122) - must print identical digests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, os, sys, warnings
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from argparse import Namespace
import numpy as np, torch
from pixelpick_amd import _lib, acquisition as acq
import formula_init as fi
L = _lib.lib()
want = os.environ["PIXELPICK_KNOBS_BUILD"] == "1"
assert _lib.knobs_build() == want and L._name.endswith("libpixelpick_hip_knobs.so" if want else "libpixelpick_hip.so"), L._name
assert hasattr(L, "pp_debug_set_x3") == want
h = hashlib.sha256()
def put(t): h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
g = torch.Generator(device="cuda").manual_seed(7)
logits = torch.randn((3, 19, 96, 160), device="cuda", generator=g) * 3
excl = (torch.rand((3, 96, 160), device="cuda", generator=g) < 0.05).to(torch.uint8)
for st in ("entropy", "least_confidence", "margin_sampling"):
    for ro in (False, True):
        for k in (20, 768):
            idx, val, m = acq.score_topk(logits, excl, st, k, return_map=True, reference_order=ro)
            put(idx); put(val); put(m)
from pixelpick_amd.networks.layers import Dropout
from pixelpick_amd.trainer import FlatTrainer
from pixelpick_amd.utils.utils import get_model
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab", weight_type="random"))
m.load_state_dict(fi.formula_state_dict(m.state_dict()))
for mod in m.modules():
    if isinstance(mod, Dropout):
        mod.p = 0.0
tr = FlatTrainer(m.cuda().train(), ignore_index=19)
x = fi.formula_input(2, 128, 192, key="rel").cuda(); y = fi.formula_labels(2, 128, 192, 19, 19, 20, key="rel").cuda()
for _ in range(3):
    put(tr.train_step(x, y))
put(tr.flat_p)
torch.cuda.synchronize()
print("DIGEST", h.hexdigest())
''' % (ROOT, ROOT)


def _run(knobs: str) -> str:
    env = dict(os.environ, PIXELPICK_KNOBS_BUILD=knobs, PIXELPICK_MNV2_WEIGHTS="random")
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST ")]
    assert len(line) == 1, out.stdout[-2000:]
    return line[0]


def test_product_library_equals_the_test_build_bit_for_bit():
    assert _run("0") == _run("1")


def test_reference_order_flag_equals_the_test_builds_process_switch():
    """PP_ACQ_REFERENCE_ORDER (per call, product ABI) selects exactly what pp_debug_set_exact_formula(1) selects process-wide in the
    test build: maps, picks and values bit-identical; and a call without the flag right after one with it is the default scorer again."""
    import numpy as np
    import torch
    from pixelpick_amd import _lib
    from pixelpick_amd import acquisition as acq
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(11)
    logits = torch.randn((2, 21, 80, 112), device="cuda", generator=g) * 3
    for st in ("entropy", "least_confidence", "margin_sampling"):
        d0 = acq.score_topk(logits, None, st, 20, return_map=True)
        f1 = acq.score_topk(logits, None, st, 20, return_map=True, reference_order=True)
        d1 = acq.score_topk(logits, None, st, 20, return_map=True)
        L.pp_debug_set_exact_formula(1)
        try:
            s1 = acq.score_topk(logits, None, st, 20, return_map=True)
        finally:
            L.pp_debug_set_exact_formula(0)
        for a, b in zip(f1, s1):
            assert torch.equal(a, b), st
        for a, b in zip(d0, d1):
            assert torch.equal(a, b), st
        if st == "entropy":
            assert not torch.equal(d0[2], f1[2])          # the two forms round differently somewhere on 17 920 pixels
            np.testing.assert_allclose(d0[2].cpu().numpy(), f1[2].cpu().numpy(), rtol=2e-5, atol=2e-6)
