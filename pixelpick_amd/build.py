"""Builds the HIP extension (hand-written HIP for gfx950) in-tree with hipcc, twice from the same sources:

    libpixelpick_hip.so        the product - exactly the C ABI of include/pixelpick_hip.h, no `pp_debug_*` symbol
    libpixelpick_hip_knobs.so  the test build - the same sources + -DPP_DEBUG_KNOBS: the planner switches of
                               include/pixelpick_hip_knobs.h and the experiment kernels behind them (tests, tools/, A/B runs)

    libpixelpick_hip_asan.so   (python -m pixelpick_amd.build --asan) the product's sources with the HOST side under AddressSanitizer
                               (-fsanitize=address -fno-gpu-sanitize): tests/test_abi_asan.py drives every entry point's argument
                               validation and the planners through it (sanitizers)

    python -m pixelpick_amd.build [--force] [--release-only] [--asan]

The .so files are git-ignored but travel to the GPU box with the gpurun snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpixelpick_hip.so")
OUT_KNOBS = os.path.join(HERE, "libpixelpick_hip_knobs.so")
OUT_ASAN = os.path.join(HERE, "libpixelpick_hip_asan.so")
ASAN_FLAGS = ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libasan", "-g", "-fno-omit-frame-pointer"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def asan_runtime():
    """Path of the shared AddressSanitizer runtime of the compiler that built the library (to LD_PRELOAD into a python driver)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang")
    out = subprocess.check_output([clang, "--print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build(force=False, verbose=True, knobs=True, asan=False):
    """Compile every source for the selected builds in parallel, link the libraries; returns the product's path."""
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    variants = [("_obj", [], OUT)] + ([("_obj_knobs", ["-DPP_DEBUG_KNOBS"], OUT_KNOBS)] if knobs else [])
    if asan:
        variants = [("_obj_asan", ASAN_FLAGS, OUT_ASAN)]
    procs, links = [], []
    for sub, extra, out in variants:
        os.makedirs(os.path.join(HERE, sub), exist_ok=True)
        objs, fresh = [], False
        for src in sources():
            obj = os.path.join(HERE, sub, os.path.basename(src) + ".o")
            objs.append(obj)
            if force or _stale(obj, [src] + hdrs):
                cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src, subprocess.Popen(cmd)))
                fresh = True
        links.append((out, objs, fresh))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for out, objs, fresh in links:
        if force or fresh or _stale(out, objs):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + (["-fsanitize=address", "-shared-libasan"] if out == OUT_ASAN else [])
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build(force="--force" in sys.argv, asan=True)
        print(OUT_ASAN)
    else:
        build(force="--force" in sys.argv, knobs="--release-only" not in sys.argv)
        print(OUT)
