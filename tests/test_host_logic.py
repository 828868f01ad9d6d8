"""CPU-only: C-ABI symbol coverage, host-side codec / stats / selection plumbing vs reference goldens."""
import os
import pickle as pkl
import re
import subprocess
import sys
from argparse import Namespace

import numpy as np
import pytest

from pixelpick_amd import _lib
from pixelpick_amd import query as ppq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.KNOBS_LIB_PATH)):
        from pixelpick_amd import build
        build.build(verbose=False)


def _declared(*headers):
    decl = set()
    for hdr in headers:
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        decl |= set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", txt))
    return decl


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return set(re.findall(r" T (pp_[a-z0-9_]+)", out)), out


def test_abi_exports_every_declared_symbol():
    """The PRODUCT library exports exactly include/pixelpick_hip.h and not one planner switch (no global mutable
    state behind the ABI); the TEST BUILD adds exactly include/pixelpick_hip_knobs.h."""
    _ensure_built()
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["pixelpick_hip.h", "pixelpick_hip_knobs.h"]
    pub, knobs = _declared("pixelpick_hip.h"), _declared("pixelpick_hip_knobs.h")
    assert pub and knobs and not (pub & knobs)
    assert not any(n.startswith("pp_debug_") for n in pub) and all(n.startswith("pp_debug_") for n in knobs)
    exported, raw = _exported(_lib.LIB_PATH)
    assert "pp_debug_" not in raw, "the product library exports a planner switch"
    assert pub == exported, f"header and product library differ: {pub ^ exported}"
    assert pub == set(_lib.SIGNATURES), f"ctypes table out of sync: {pub ^ set(_lib.SIGNATURES)}"
    exported_k, _ = _exported(_lib.KNOBS_LIB_PATH)
    assert exported_k == pub | knobs, f"test build: {exported_k ^ (pub | knobs)}"
    assert knobs == set(_lib.KNOB_SIGNATURES), f"ctypes knob table out of sync: {knobs ^ set(_lib.KNOB_SIGNATURES)}"
    L = _lib.lib()           # loads without a GPU (the suite runs on the test build: tests/conftest.py)
    assert _lib.knobs_build() and L.pp_version() >= 100


def test_product_library_loads_and_validates_without_the_switches():
    """A process of its own on libpixelpick_hip.so: every public symbol binds, no pp_debug_* attribute exists, argument validation and
    the per-call PP_ACQ_REFERENCE_ORDER flag are accepted (a flag outside the enum is not)."""
    _ensure_built()
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "from pixelpick_amd import _lib\n"
        "L = _lib.lib()\n"
        "assert not _lib.knobs_build() and L._name.endswith('libpixelpick_hip.so')\n"
        "assert not hasattr(L, 'pp_debug_set_x3')\n"
        "for s in (0, 1, 2, 0x100, 0x102):\n"
        "    assert L.pp_acq_score_topk(None, 1, 19, 4, 4, 0, 0, 0, 0, None, s, 5, None, None, None, None, 0, None) == -1 and b'null' in L.pp_last_error()\n"
        "assert L.pp_acq_score_topk(1, 1, 19, 4, 4, 0, 0, 0, 0, None, 3, 5, None, None, None, None, 0, None) == -1 and b'strategy' in L.pp_last_error()\n"
        "assert L.pp_acq_score_topk(1, 1, 19, 4, 4, 0, 0, 0, 0, None, 0x200, 5, None, None, None, None, 0, None) == -1\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ, PIXELPICK_KNOBS_BUILD="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_abi_argument_validation_without_gpu():
    _ensure_built()
    L = _lib.lib()
    assert L.pp_acq_score_topk(None, 1, 19, 4, 4, 0, 0, 0, 0, None, 0, 5, None, None, None, None, 0, None) == -1
    assert b"null" in L.pp_last_error()
    assert L.pp_acq_workspace_bytes(256, 19, 256, 512, 20) > 0
    assert L.pp_acq_workspace_bytes(1, 19, 256, 512, 6553) >= 256 * 512 * 4
    assert L.pp_topk_select(None, 1, 10, 3, 1, None, None, None, 0, None) == -1


def test_acquisition_workspace_sizes_cover_every_large_k_path():
    """pp_acq_workspace_bytes is the one place a caller learns how much scratch the large-k selection needs (host arithmetic only).  It must
    hold a score map (4 B / pixel) in every case - the map path, and the exact fallback of the list select, write one - and where the list
    select is offered (k <= N / 8, N >= 16384) the per-wave candidate segments (8 B per pixel + the pad) plus the per-image keys and per-wave
    counts; it grows with B and is 0 for arguments no launch would accept (instead of wrapping around)."""
    L = _lib.lib()
    for B, H, W, k in ((256, 256, 512, 6553), (8, 1024, 1024, 5000), (3, 128, 160, 1024), (2, 64, 96, 307), (1, 100, 172, 860)):
        N = H * W
        ws = L.pp_acq_workspace_bytes(B, 19, H, W, k)
        topk = L.pp_topk_workspace_bytes(B, N, k)
        assert ws >= B * N * 4 + topk
        if k * 8 <= N and N >= 16384:
            segs = -(-N // 2048) * 8                    # 256-pixel segments
            assert ws >= B * segs * (256 + 32) * 8 + topk + B * 4 + B * segs * 4
        assert L.pp_acq_workspace_bytes(2 * B, 19, H, W, k) > ws
    assert L.pp_acq_workspace_bytes(1, 19, 1 << 16, 1 << 16, 100) == 0          # H * W beyond int32
    assert L.pp_acq_workspace_bytes(1, 19, 64, 64, 64 * 64 + 1) == 0             # k > H * W
    assert L.pp_topk_workspace_bytes(4, 131072, 1 << 31) == 0


def test_launch_plan_executor_knows_every_enqueuing_entry_point():
    """csrc/plan.hip holds one typed thunk per entry point that enqueues work; its arity comes from the function's own prototype
    and must agree with the ctypes table.  The plan API itself (create / add / host break / replay of an empty stretch) needs no GPU."""
    import ctypes
    _ensure_built()
    L = _lib.lib()
    for name, (_, args) in _lib.SIGNATURES.items():
        fn = getattr(L, name)
        got = L.pp_plan_entry_args(ctypes.cast(getattr(fn, "__wrapped__", fn), ctypes.c_void_p))
        assert got == (len(args) if _lib._is_launch(name) else -1), name
    h = L.pp_plan_create()
    try:
        slots = (ctypes.c_uint64 * 3)(1, 2, 3)
        assert L.pp_plan_add_call(h, ctypes.cast(L.pp_version, ctypes.c_void_p), slots, 0) == -4     # not an enqueuing entry
        assert L.pp_plan_add_call(h, ctypes.cast(L.pp_add2d, ctypes.c_void_p), slots, 3) == -1       # wrong arity
        assert b"9 arguments" in L.pp_last_error()
        assert L.pp_plan_add_host_break(h) == 0 and L.pp_plan_add_host_break(h) == 0 and L.pp_plan_size(h) == 2
        nxt = ctypes.c_int64(-1)
        assert L.pp_plan_replay(h, 0, ctypes.byref(nxt)) == 0 and nxt.value == 1
        assert L.pp_plan_replay(h, 1, ctypes.byref(nxt)) == 0 and nxt.value == 2
        assert L.pp_plan_replay(h, 2, ctypes.byref(nxt)) == 0 and nxt.value == 2
    finally:
        L.pp_plan_destroy(h)
    # slot packing of the binding: float in the low four bytes, None pointer = 0, negative integers sign-extended
    assert _lib._slot(ctypes.c_float(1.0)) == 0x3F800000
    assert _lib._slot(ctypes.c_void_p(None)) == 0
    assert _lib._slot(ctypes.c_int(-1)) == 0xFFFFFFFFFFFFFFFF


def test_debug_knobs_invalidate_the_engine_memo_tables():
    """plan-dependent answers memoised on the host (workspace sizes, `_ok` queries) are refilled after any
    pp_debug_set_* knob moved."""
    _ensure_built()
    from pixelpick_amd import engine as E
    L = _lib.lib()
    E._wsbytes("pp_conv2d_fwd_workspace_bytes", 4, 16, 32, 960, 160, 1, 1, 1, 0, 1)
    assert E._WS_BYTES
    L.pp_debug_set_x3(1)
    E._memo_epoch()
    assert not E._WS_BYTES and not E._CONV_WS_BYTES and not E._ACCEPTS_AFFINE


def test_shard_dataloader_declines_samplers_it_cannot_restate():
    """subset / weighted / user samplers fall back to enumerate-and-skip instead of silently becoming a sequential pass."""
    import torch
    from torch.utils.data import DataLoader, SubsetRandomSampler, TensorDataset, WeightedRandomSampler
    from pixelpick_amd import dist_utils
    ds = TensorDataset(torch.arange(10))
    assert dist_utils.shard_dataloader(DataLoader(ds, batch_size=2, sampler=SubsetRandomSampler([1, 2, 3])), 0, 2, False) is None
    assert dist_utils.shard_dataloader(DataLoader(ds, batch_size=2, sampler=WeightedRandomSampler([1.0] * 10, 4)), 0, 2, False) is None
    g = torch.Generator()
    dl = dist_utils.shard_dataloader(DataLoader(ds, batch_size=2, shuffle=True, generator=g, timeout=0), 1, 2, True, seed=3)
    assert dl is not None and dl.generator is g and len(dl) == 2
    assert dist_utils.augment_seed(3, 0) != dist_utils.augment_seed(3, 1)


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpixelpick_hip.so")
    monkeypatch.setattr(_lib, "KNOBS_LIB_PATH", "/nonexistent/libpixelpick_hip_knobs.so")
    with pytest.raises(_lib.PixelPickHipError):
        _lib.lib()


def test_cpu_tensor_is_rejected_not_silently_computed():
    import torch
    from pixelpick_amd import acquisition as acq
    with pytest.raises(_lib.PixelPickHipError):
        acq.score_topk(torch.zeros(1, 19, 4, 4), None, "entropy", 2)


def test_codec_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "acq_codec.npz"))
    enc = ppq.QuerySelector.encode_query("a/b.png", g["mask"].shape, g["mask"])
    info = enc["a/b.png"]
    assert info["height"], info["width"] == g["mask"].shape
    assert info["x_coords"].dtype == np.int64
    np.testing.assert_array_equal(info["x_coords"], g["enc_x"])
    np.testing.assert_array_equal(info["y_coords"], g["enc_y"])
    dec = ppq.QuerySelector.decode_queries(enc)
    assert len(dec) == 1 and dec[0].dtype == np.bool_
    np.testing.assert_array_equal(dec[0], g["dec_single"])
    enc2 = {"z.png": dict(info), "a.png": dict(info)}
    enc2["z.png"]["category_id"] = g["cat"].tolist()
    d2 = ppq.QuerySelector.decode_queries(enc2, ignore_index=11, return_as_dict=True)
    assert d2["z.png"].dtype == np.int64
    np.testing.assert_array_equal(d2["z.png"], g["dec_cat_z"])
    np.testing.assert_array_equal(d2["a.png"], g["dec_plain_a"])
    l2 = ppq.QuerySelector.decode_queries(enc2, ignore_index=11)
    assert (l2[0].dtype == np.bool_) == bool(g["dec_list_order_first_is_a"])
    with pytest.raises(ValueError):
        ppq.QuerySelector.decode_queries({})


def test_merge_previous_query_files_matches_reference(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "acq_codec.npz"))
    h, w = g["merge_img0"].shape
    files = []
    for r in range(3):
        d = tmp_path / f"{r}_query"
        d.mkdir()
        e = {}
        for tag in ("img0", "img1"):
            key = f"merge_in_{r}_{tag}_x"
            if key in g:
                e[f"{tag}.png"] = {"height": h, "width": w, "x_coords": g[key], "y_coords": g[f"merge_in_{r}_{tag}_y"],
                                   "category_id": g[f"merge_in_{r}_{tag}_c"].tolist()}
        with open(d / "queries.pkl", "wb") as f:
            pkl.dump(e, f)
        files.append(str(d / "queries.pkl"))
    assert sorted(ppq.gather_previous_query_files(str(tmp_path))) == sorted(files)
    merged = ppq.merge_previous_query_files(files, ignore_index=11, verbose=False)
    np.testing.assert_array_equal(merged["img0.png"], g["merge_img0"])
    np.testing.assert_array_equal(merged["img1.png"], g["merge_img1"])


def _args(**kw):
    base = dict(dataset_name="cs", debug=False, dir_root="/tmp", experim_name="t", ignore_index=19, mc_n_steps=20,
                n_classes=19, n_pixels_by_us=10, network_name="deeplab", query_strategy="entropy",
                reverse_order=False, stride_total=8, top_n_percent=0.0, use_mc_dropout=False, vote_type="hard")
    base.update(kw)
    return Namespace(**base)


@pytest.mark.parametrize("st", ["entropy", "least_confidence", "margin_sampling"])
def test_top_percent_subsample_host_step_matches_reference(golden_dir, st):
    """query.py:63-68 host RNG step, given the reference's value-sorted top-5% order."""
    g = np.load(os.path.join(golden_dir, "acq_select_modes.npz"))
    h, w = g[f"{st}_uc"].shape
    qs = ppq.QuerySelector(_args(query_strategy=st, top_n_percent=0.05), None, device="cpu")
    np.random.seed(int(g["np_seed_top5"]))
    q = qs._finish_selection(g[f"{st}_top5_order"], h, w)
    assert np.flatnonzero(q.reshape(-1)).tolist() == g[f"{st}_top5_sel"].tolist()


def test_query_stats_host_part_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "acq_end_to_end.npz"))
    st = "entropy"
    qs = ppq.QueryStats(_args())
    ys = g[f"{st}_ys"]
    for i in range(ys.shape[0]):
        q = np.zeros(ys[i].shape, dtype=bool)
        q[g[f"{st}_y_{i}"], g[f"{st}_x_{i}"]] = True
        qs.update_from_picked(q, ys[i], [0.0] * int(q.sum()))
    cnt = np.array([qs.dict_label_cnt[l] for l in range(19)])
    np.testing.assert_array_equal(cnt, g[f"{st}_stats_label_cnt"])
    assert np.isclose(np.mean(qs.list_n_unique_labels), float(g[f"{st}_stats_avg_n_unique"]))
    assert np.isclose(np.mean(qs.list_spatial_coverage), float(g[f"{st}_stats_avg_cov"]))


def test_exclusion_mask_is_reinterpreted_not_converted():
    """acquisition._exclude_u8: bool masks (numpy or torch) become uint8 by a zero-copy view; other dtypes by != 0."""
    import torch
    from pixelpick_amd import acquisition as acq
    rng = np.random.RandomState(0)
    m = rng.rand(2, 5, 7) < 0.3
    for src in (m, torch.from_numpy(m), torch.from_numpy(m.astype(np.int64) * 3), m[:, ::-1, :], torch.from_numpy(m.astype(np.uint8))):
        want = np.asarray(src if isinstance(src, np.ndarray) else src.numpy()) != 0
        got = acq._exclude_u8(src, 2, 5, 7, torch.device("cpu"))
        assert got.dtype == torch.uint8 and tuple(got.shape) == (2, 5, 7)
        np.testing.assert_array_equal(got.numpy(), want.astype(np.uint8))
    assert acq._exclude_u8(None, 2, 5, 7, torch.device("cpu")) is None


def test_numpy_choice_over_an_array_equals_choice_over_its_positions():
    """The pipelined QuerySelector draws POSITIONS of the value-sorted candidate list instead of the candidates themselves
    (query.py:63-64 `np.random.choice(ind, n, False)`): same picks, same RNG consumption."""
    import numpy as np
    for s in range(6):
        a = np.random.RandomState(100 + s).permutation(200000)[:6553]
        np.random.seed(s)
        r1 = np.random.choice(a, 10, False)
        nxt1 = np.random.rand()
        np.random.seed(s)
        pos = np.random.choice(len(a), 10, False)
        nxt2 = np.random.rand()
        assert (a[pos] == r1).all() and nxt1 == nxt2


def test_launch_plan_records_and_replays_in_order():
    """_lib.LaunchPlan mechanics without a GPU: launches are recorded with converted arguments, queries (workspace sizes,
    debug knobs) are not, plan_note() callables ride in order, a failing call is reported on replay."""
    import ctypes
    from pixelpick_amd import _lib

    log = []

    class _Fn:
        def __init__(self, name, argtypes, rc=0):
            self.__name__, self.argtypes, self.rc = name, argtypes, rc

        def __call__(self, *args):
            log.append((self.__name__, tuple(getattr(a, "value", a) for a in args)))
            return self.rc

    class _Fake:
        pp_add2d = _Fn("pp_add2d", [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float])
        pp_topk_workspace_bytes = _Fn("pp_topk_workspace_bytes", [ctypes.c_int64])
        pp_bad = _Fn("pp_bad", [ctypes.c_int], rc=-3)

        @staticmethod
        def pp_last_error():
            return b"boom"

    plan = _lib.LaunchPlan()
    rec = _lib._RecordingLib(_Fake, plan)
    _lib._recording[0] = rec
    try:
        assert _lib.recording()
        rec.pp_add2d(None, 7, 0.5)
        rec.pp_topk_workspace_bytes(3)                 # a query: executed, not recorded
        _lib.plan_note(log.append, "note")
        rec.pp_add2d(1234, 8, 1.5)
        assert rec.pp_bad(1) == -3                     # failed launches are not recorded
    finally:
        _lib._recording[0] = None
    assert len(plan) == 3
    assert all(isinstance(a, ctypes._SimpleCData) for a in plan.calls[0][1])
    first = list(log)
    del log[:]
    plan.replay()
    assert log == [("pp_add2d", (None, 7, 0.5)), "note", ("pp_add2d", (1234, 8, 1.5))]
    assert [e for e in first if e != ("pp_topk_workspace_bytes", (3,)) and e != ("pp_bad", (1,))] == log
    plan.calls.append((_Fake.pp_bad, (ctypes.c_int(1),)))
    with pytest.raises(_lib.PixelPickHipError):
        plan.replay()


def test_cv2_fixed_point_gaussian_taps():
    """oracle/augment.py's restatement of OpenCV's getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED and the product's own
    builder (pixelpick_amd/augment.py) agree; taps are symmetric, sum to exactly 256 (8.8 fixed point), a sigma far below one pixel
    is the identity, and a constant image stays constant through the integer passes."""
    from oracle import augment as oa
    from pixelpick_amd.augment import cv2_gaussian_kernel_q8
    for ks in (1, 3, 5, 7, 25, 33):
        for sigma in (0.1, 0.37, 0.83, 1.2, 1.97, 5.0):
            a, b = oa.gaussian_kernel_q8(ks, sigma), cv2_gaussian_kernel_q8(ks, sigma)
            assert np.array_equal(a, b.astype(np.int64)) and a.sum() == 256 and np.array_equal(a, a[::-1]) and (a >= 0).all()
    assert oa.gaussian_kernel_q8(25, 0.1)[12] == 256
    assert oa.gaussian_kernel_q8(3, 0.8).tolist() == [61, 134, 61]          # exp(-0.78125) / (1 + 2 exp(-0.78125)) * 256 = 61.2
    x = np.full((9, 11, 3), 137, np.uint8)
    assert (oa.gaussian_blur(x, 7, 1.3) == 137).all()
    rng = np.random.RandomState(0)
    x = rng.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    assert np.abs(oa.gaussian_blur(x, 7, 1.2).astype(int) - oa.gaussian_blur_float(x, 7, 1.2).astype(int)).max() <= 1
