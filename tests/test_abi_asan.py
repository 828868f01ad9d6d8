"""SURVEY.md 5 (sanitizers) / 8(b): the host side of the C ABI under AddressSanitizer.  libpixelpick_hip_asan.so = the product's sources built
with -fsanitize=address -fno-gpu-sanitize (python -m pixelpick_amd.build --asan; built here when missing: ~3.5 minutes once).  A python of its
own (the ASan runtime preloaded, no torch, no GPU) calls every entry point with null / hostile / well-formed arguments: tests/abi_asan_driver.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_entry_points_host_side_is_clean_under_address_sanitizer():
    if os.path.exists("/dev/kfd"):
        pytest.skip("the driver hands fake device pointers to the launchers: a box without a GPU only")
    from pixelpick_amd import build
    if build._stale(build.OUT_ASAN, build.sources() + [os.path.join(ROOT, "include", "pixelpick_hip.h")]):
        build.build(verbose=False, asan=True)
    rt = build.asan_runtime()
    assert rt, "the compiler's shared ASan runtime was not found"
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_asan_driver.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "asan driver ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
