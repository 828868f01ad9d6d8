#!/usr/bin/env python3
"""Writes tests/golden/aug_blur_cv2.npz: cv2.GaussianBlur itself on seeded 8-bit images, for every kernel size the reference
can ask for and a spread of sigmas (datasets/base_dataset.py:140,192-208: ksize = int(0.1 * smaller_crop_side // 2 * 2 + 1),
sigma uniform in [0.1, 2.0)).

This closes the one "parity unpinned" op of the device data path (oracle/augment.py:gaussian_blur restates OpenCV's 8-bit
fixed-point path from its sources; this image has no cv2, so no fixture written by the real library exists yet).

    pip install opencv-python-headless numpy      # any box with network access; OpenCV >= 3.4.2
    python tools/gen_golden_blur_cv2.py           # writes tests/golden/aug_blur_cv2.npz (~200 KB), prints the cv2 version

Once the file exists, tests/test_oracle_golden.py::test_blur_oracle_equals_cv2_fixture (CPU: oracle == cv2) and
tests/test_augment_gpu.py::test_blur_kernel_equals_cv2_fixture (GPU: pp_aug_blur_q8 == cv2) pick it up and HARD-FAIL on any
differing pixel; while it is absent they skip with the reason "no cv2-written fixture".  The images are regenerated from the
seeds stored in the file, so the fixture holds only parameters and cv2's outputs.
"""
import os
import sys

import numpy as np

KSIZES = (3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25)          # 25 = the Cityscapes / CamVid crops' value; 31 for the 320 x 320 VOC crop
KSIZES_EXTRA = (31,)
SIGMAS = (0.1, 0.25, 0.5, 0.8, 1.0, 1.3, 1.7, 1.9999)
SHAPE = (48, 80, 3)


def image(seed: int) -> np.ndarray:
    """The test image of a case: noise + a ramp + a few saturated blocks (exercises the rounding and the reflect-101 border)."""
    rng = np.random.RandomState(seed)
    a = rng.randint(0, 256, SHAPE).astype(np.int64)
    a[:, :, 1] = (a[:, :, 1] // 4 + np.arange(SHAPE[1])[None, :] * 2) % 256
    a[4:12, 6:20] = 255
    a[30:40, 50:70] = 0
    return a.astype(np.uint8)


def main():
    import cv2
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "aug_blur_cv2.npz")
    ks, sg, seeds, res = [], [], [], []
    n = 0
    for k in KSIZES + KSIZES_EXTRA:
        for s in SIGMAS:
            seed = 1000 + n
            img = image(seed)
            res.append(cv2.GaussianBlur(img, (k, k), s))
            ks.append(k); sg.append(s); seeds.append(seed)
            n += 1
    np.savez_compressed(out, ksize=np.array(ks, np.int32), sigma=np.array(sg, np.float64), seed=np.array(seeds, np.int64),
                        blurred=np.stack(res), cv2_version=np.array(cv2.__version__))
    print(f"wrote {os.path.normpath(out)}: {n} cases, cv2 {cv2.__version__}")


if __name__ == "__main__":
    sys.exit(main())
