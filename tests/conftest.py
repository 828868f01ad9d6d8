import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# DeepLab starts from the ImageNet MobileNetV2 file unless a random backbone is asked for (networks/mobilenet_v2.py); the tests
# fill every weight themselves (formula / seeds), so they ask for it once here.  tests/test_checkpoint_format_gpu.py overrides it.
os.environ.setdefault("PIXELPICK_MNV2_WEIGHTS", "random")
# The suite runs on the TEST BUILD (libpixelpick_hip_knobs.so = the product's sources + -DPP_DEBUG_KNOBS): the parity tests force
# kernel forms through the pp_debug_* planner switches, which the product library does not export.  tests/test_release_build_gpu.py
# holds the product library to the same results in a process of its own.
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
