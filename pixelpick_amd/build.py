"""Builds libpixelpick_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m pixelpick_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpixelpick_hip.so")
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


def build(force=False, verbose=True):
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(HERE, "_obj", os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
