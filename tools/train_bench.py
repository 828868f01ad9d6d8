#!/usr/bin/env python3
"""Quick timing of the DeepLabv3+-MNv2 (NET=FPN: FPNSeg-ResNet50) train step (FlatTrainer) at the BASELINE config: B=4, 256x512, C=19."""
import contextlib, os, sys, time, json, warnings
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd.utils.utils import get_model
from pixelpick_amd.trainer import FlatTrainer

def main():
    B, H, W, C = int(os.environ.get("B", 4)), 256, 512, 19
    steps, warm = int(os.environ.get("STEPS", 10)), 3
    args = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=os.environ.get("NET", "deeplab"), weight_type="random",
                     n_layers=50, use_softmax=True, use_dilated_resnet=True, width_multiplier=1.0)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(args).cuda().train()
    if os.environ.get("GEMMPW"):                      # pp_debug_set_gemm_pw word (0: the pointwise GEMM kernel off)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_gemm_pw(int(os.environ["GEMMPW"], 0))
    if os.environ.get("X3F"):                         # pp_debug_set_x3f word (A/B of the in-kernel activation split, csrc/conv_x3f.hip)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_x3f(int(os.environ["X3F"], 0))
    if os.environ.get("X3MODE"):                      # pp_debug_set_x3 word (A/B of the bf16x3 kernels)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_x3(int(os.environ["X3MODE"]))
    if os.environ.get("NO_WGRAD"):                    # TIMING ONLY: no convolution weight gradients at all (what the second queue costs the step)
        for p_ in m.parameters():
            if p_.dim() == 4:
                p_.requires_grad_(False)
    if os.environ.get("CONVVAR"):                     # pp_debug_set_conv_variant word
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_conv_variant(int(os.environ["CONVVAR"]))
    if os.environ.get("BNBYTES"):                     # pp_debug_set_bn_bytes_per_block word (-1: no row cache, -2: no reversed second pass)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_bn_bytes_per_block(int(os.environ["BNBYTES"]))
    if os.environ.get("BNTARGET"):                    # pp_debug_set_bn_target: blocks the single-launch BatchNorm kernels aim at (0 = one per CU)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_bn_target(int(os.environ["BNTARGET"]))
    if os.environ.get("ROWS"):                        # pp_debug_set_conv_rows bits (whole-row kernels of the narrow pointwise layers)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_conv_rows(int(os.environ["ROWS"]))
    if os.environ.get("BNFUSE"):                      # pp_debug_set_conv_bn_fuse bits (which fused conv + BatchNorm launches are offered)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_conv_bn_fuse(int(os.environ["BNFUSE"]))
    if os.environ.get("WGRAD_TARGET"):                # pp_debug_set_wgrad_target word (blocks the weight-gradient kernels aim at)
        from pixelpick_amd import _lib
        _lib.lib().pp_debug_set_wgrad_target(int(os.environ["WGRAD_TARGET"]))
    tr = FlatTrainer(m, ignore_index=C)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, 3, H, W, device="cuda", generator=g)
    y = torch.full((B, H, W), C, dtype=torch.int64, device="cuda")
    for b in range(B):
        idx = torch.randperm(H * W, device="cuda", generator=g)[:20]
        y[b].view(-1)[idx] = torch.randint(0, C, (20,), device="cuda", generator=g)
    mp = os.environ.get("MAIN_PRIORITY")              # run the step on a stream of this HIP priority (default: the current stream)
    ctx = torch.cuda.stream(torch.cuda.Stream(priority=int(mp))) if mp is not None else contextlib.nullcontext()
    with ctx:
        if os.environ.get("REPLAY"):                  # launch-plan replay (FlatTrainer.enable_replay) instead of eager steps
            tr.enable_replay(x, y, warmup=1)
        for _ in range(warm):
            loss = tr.train_step(x, y)
        torch.cuda.synchronize()
        if os.environ.get("HOSTPROF"):                # cProfile of the host side of HOSTPROF steps, each issued into empty queues
            import cProfile, pstats
            pr = cProfile.Profile()
            for _ in range(int(os.environ["HOSTPROF"])):
                torch.cuda.synchronize()
                pr.enable()
                tr.train_step(x, y)
                pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("tottime").print_stats(45)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.train_step(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"ms_per_step": dt * 1e3, "img_per_s": B / dt, "loss": loss.item()}))

if __name__ == "__main__":
    main()
