"""Driver of tests/test_abi_asan.py: runs in a python of its own with the AddressSanitizer runtime preloaded, WITHOUT torch (the binding table
is read from pixelpick_amd/_lib.py with a stub in torch's place) and without a GPU.  Every entry point of include/pixelpick_hip.h is called
through libpixelpick_hip_asan.so with (1) null / zero arguments, (2) hostile sizes in the pure queries, (3) well-formed small and BASELINE-size
shapes with non-null (never dereferenced on the host) pointers, so that argument validation, the tap tables, the planners and the launch-plan
executor run on the host side under the sanitizer; without a device every launch ends in PP_ERR_LAUNCH.  Any invalid host access aborts the
process (ASan), which the test sees as a non-zero exit."""
import ctypes
import importlib.util
import itertools
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_table():
    sys.modules["torch"] = types.ModuleType("torch")               # _lib.py imports torch only to preload its HIP runtime
    spec = importlib.util.spec_from_file_location("pp_lib_table", os.path.join(ROOT, "pixelpick_amd", "_lib.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.SIGNATURES


def main():
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)        # a hang (a planner looping on a hostile size) must end the driver, not the suite
    assert not os.path.exists("/dev/kfd") or os.environ.get("PP_ASAN_ALLOW_GPU") == "1", "meant for a box without a GPU (fake device pointers)"
    sigs = load_table()
    L = ctypes.CDLL(os.path.join(ROOT, "pixelpick_amd", "libpixelpick_hip_asan.so"))
    fns = {}
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
        fns[name] = fn
    calls = 0
    host_ptr_fns = {"pp_aug_to_tensor", "pp_wgrad_reduce_batch", "pp_plan_add_call", "pp_plan_replay", "pp_plan_destroy", "pp_plan_size",
                    "pp_plan_add_event_record", "pp_plan_add_stream_wait", "pp_plan_add_join", "pp_plan_add_host_break", "pp_plan_entry_args",
                    "pp_set_kernel_events", "pp_set_comm_cu_reserve"}
    # (1) all-null / all-zero
    for name, (res, args) in sigs.items():
        if name in host_ptr_fns or name in ("pp_last_error", "pp_version", "pp_plan_create"):
            continue
        z = [t() if t not in (ctypes.c_void_p,) else None for t in args]
        z = [0 if isinstance(v, (ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint64)) else (0.0 if isinstance(v, ctypes.c_float) else v) for v in z]
        rc = fns[name](*z)
        calls += 1
        if res is ctypes.c_int and not name.endswith(("_ok", "_accepts_affine_in", "_capacity", "_rows_cached", "_reserve")):
            assert rc < 0, (name, rc)
            assert L.pp_last_error(), name
    # (2) hostile sizes in the pure queries
    big = [0, 1, -1, 3, 7, 49, 50, 2**31 - 1, -2**31]
    for name, (res, args) in sigs.items():
        if not name.endswith(("_bytes", "_rows", "_ints", "_ok", "_accepts_affine_in", "_rows_cached")):
            continue
        n = len(args)
        for v in big:
            fns[name](*[v] * n)
            calls += 1
        for combo in itertools.islice(itertools.product([1, 4, 64, 2**20, 2**31 - 1], repeat=min(n, 4)), 200):
            a = (list(combo) + [3] * n)[:n]
            fns[name](*a)
            calls += 1
    # (3) planners and tap tables on well-formed shapes; pointers are fake DEVICE addresses, never read on the host
    P = 0x7F0000000000
    conv_shapes = [(4, 64, 128, 304, 256, 3, 3, 1, 1, 1), (4, 16, 32, 320, 256, 3, 3, 1, 18, 18), (4, 16, 32, 320, 256, 3, 3, 1, 6, 6),
                   (4, 32, 64, 256, 1024, 1, 1, 1, 0, 1), (4, 32, 64, 1024, 256, 1, 1, 1, 0, 1), (4, 256, 512, 3, 32, 3, 3, 2, 1, 1),
                   (4, 128, 256, 3, 64, 7, 7, 2, 3, 3), (2, 9, 11, 24, 144, 1, 1, 1, 0, 1), (4, 16, 32, 960, 320, 1, 1, 1, 0, 1),
                   (1, 5, 5, 8, 8, 7, 7, 1, 3, 1), (4, 64, 64, 256, 128, 1, 1, 2, 0, 1), (2, 33, 31, 20, 12, 3, 3, 2, 1, 1), (1, 1, 1, 4, 4, 3, 3, 1, 1, 1)]
    ws = 1 << 30
    for (B, H, W, ci, co, kh, kw, s, p, d) in conv_shapes:
        Ho, Wo = (H + 2 * p - d * (kh - 1) - 1) // s + 1, (W + 2 * p - d * (kw - 1) - 1) // s + 1
        if Ho < 1 or Wo < 1:
            continue
        for q in ("pp_conv2d_fwd_workspace_bytes", "pp_conv2d_bwd_data_workspace_bytes", "pp_conv2d_bwd_weight_workspace_bytes",
                  "pp_conv2d_fwd_stats_rows", "pp_conv2d_fwd_bn_train_ok", "pp_conv2d_bwd_data_bn_bwd_ok", "pp_conv2d_fwd_accepts_affine_in",
                  "pp_conv2d_fwd_bn_train_xchg_bytes", "pp_conv2d_bwd_data_bn_bwd_xchg_bytes"):
            fns[q](B, H, W, ci, co, kh, kw, s, p, d)
        for which in range(5):
            fns["pp_conv2d_x3_planes_bytes"](which, B, H, W, ci, co, kh, kw, s, p, d)
        rc = fns["pp_conv2d_fwd"](P, ci, B, H, W, ci, P, None, kh, kw, s, p, d, P, co, co, P, ws, None)
        assert rc in (0, -5, -4, -1), ("fwd", rc, L.pp_last_error())
        rc = fns["pp_conv2d_bwd_data"](P, co, B, Ho, Wo, co, P, kh, kw, s, p, d, P, ci, H, W, ci, 0, P, ws, None)
        assert rc in (0, -5, -4, -1), ("bwd_data", rc, L.pp_last_error())
        rc = fns["pp_conv2d_bwd_weight"](P, ci, B, H, W, ci, P, co, co, kh, kw, s, p, d, P, None, P, ws, None)
        assert rc in (0, -5, -4, -1, -3), ("bwd_weight", rc, L.pp_last_error())
        calls += 3
    for (B, C, H, W, k) in [(1, 19, 4, 4, 5), (256, 19, 256, 512, 20), (8, 19, 1024, 2048, 20), (2, 150, 33, 17, 6), (1, 21, 320, 320, 5120), (3, 11, 360, 480, 8640)]:
        nb = fns["pp_acq_workspace_bytes"](B, C, H, W, k)
        for st in (0, 1, 2, 0x100, 0x102):
            rc = fns["pp_acq_score_topk"](P, B, C, H, W, C * H * W, H * W, W, 1, P, st, k, P, P, P, P, nb, None)
            assert rc in (0, -5), ("acq", rc, L.pp_last_error())
            rc = fns["pp_acq_score_topk"](P, B, C, H, W, C * H * W, 1, W * C, C, None, st, k, P, None, None, P, nb, None)      # channels-last strides
            assert rc in (0, -5), ("acq nhwc", rc, L.pp_last_error())
            calls += 2
        assert fns["pp_acq_score_topk"](P, B, C, H, W, C * H * W, H * W, W, 1, P, 0, H * W + 1, P, P, P, P, nb, None) == -2      # k > H*W
        assert fns["pp_acq_score_topk"](P, B, C, H, W, C * H * W, H * W, W, 1, P, 0, k, P, P, P, P, max(nb - 1, 0), None) == -3  # workspace
        nbl = fns["pp_acq_lowres_workspace_bytes"](B, C, H, W, k)
        rc = fns["pp_acq_lowres_score_topk"](P, C, B, C, max(H // 4, 1), max(W // 4, 1), H, W, 1, H, W, P, 0, k, P, P, None, P, nbl, None)
        assert rc in (0, -5), ("lowres", rc, L.pp_last_error())
        nbt = fns["pp_topk_workspace_bytes"](B, H * W, k)
        assert fns["pp_topk_select"](P, B, H * W, k, 1, P, P, P, nbt, None) in (0, -5)
    # the launch-plan executor: host-only paths
    h = fns["pp_plan_create"]()
    slots = (ctypes.c_uint64 * 32)(*range(32))
    assert fns["pp_plan_add_call"](h, ctypes.cast(L.pp_version, ctypes.c_void_p), slots, 0) == -4
    assert fns["pp_plan_add_call"](h, ctypes.cast(L.pp_add2d, ctypes.c_void_p), slots, 3) == -1
    assert fns["pp_plan_add_call"](h, ctypes.cast(L.pp_add2d, ctypes.c_void_p), slots, 9) == 0
    assert fns["pp_plan_add_host_break"](h) == 0
    nxt = ctypes.c_int64(-1)
    fns["pp_plan_replay"](h, 0, ctypes.byref(nxt))
    fns["pp_plan_replay"](h, 1, ctypes.byref(nxt))
    assert fns["pp_plan_size"](h) == 2
    fns["pp_plan_destroy"](h)
    fns["pp_plan_destroy"](None)
    print(f"asan driver ok: {calls} calls")


if __name__ == "__main__":
    main()
