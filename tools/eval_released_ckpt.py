#!/usr/bin/env python3
"""Released-checkpoint leg of SURVEY.md 8(f)3 (README.md:113-118 of the reference): download the five published PixelPick
checkpoints, load each through THIS package's `load_state_dict`, evaluate on the dataset's validation split on the GPU and print
the mean IoU beside the README's figure.  Needs what this container lacks: network access, the datasets, torchvision-free PIL
loading (below) - so it is committed ready to run, not run.

    python tools/eval_released_ckpt.py --root /data --only cs_dl            # one model
    python tools/eval_released_ckpt.py --root /data --download-only         # just fetch + unzip into <root>/pixelpick_ckpt/

Dataset layout expected under --root (the reference's own, datasets/{cityscapes,camvid,voc}.py):
    cityscapes/leftImg8bit/val/*/*.png + gtFine/val/*/*_labelIds.png ; camvid/val + valannot ; VOCdevkit/VOC2012/{JPEGImages,SegmentationClass,ImageSets/Segmentation/val.txt}
A result within 0.5 mIoU of the README figure (its own run-to-run spread, README.md:85-109) closes the row; the script exits
non-zero otherwise.
"""
import argparse
import io
import os
import sys
import urllib.request
import zipfile
from argparse import Namespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BASE = "https://www.robots.ox.ac.uk/~vgg/research/pixelpick/shared_files/"
# key -> (zip, dataset, network_name, n_classes, ignore_index, README mIoU)                     README.md:113-118
MODELS = {
    "cv_dl": ("cv_dl_margin_sampling_100ppi.zip", "cv", "deeplab", 11, 11, 56.1),
    "cs_dl": ("cs_dl_margin_sampling_100ppi.zip", "cs", "deeplab", 19, 19, 56.8),
    "cs_fpn50": ("cs_fpn50_margin_sampling_100ppi.zip", "cs", "FPN", 19, 19, 63.3),
    "voc_dl": ("voc_dl_margin_sampling_50ppi.zip", "voc", "deeplab", 21, 255, 57.4),
    "voc_fpn50": ("voc_fpn50_margin_sampling_50ppi.zip", "voc", "FPN", 21, 255, 68.0),
}
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
# Cityscapes labelIds -> 19 train ids (datasets/cityscapes.py _cityscapes_classes_to_labels; 19 = void)
CS_MAP = np.full(256, 19, np.uint8)
for _train, _lid in enumerate((7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33)):
    CS_MAP[_lid] = _train


def fetch(key: str, dst: str) -> str:
    name = MODELS[key][0]
    d = os.path.join(dst, name[:-4])
    if not os.path.isdir(d):
        os.makedirs(dst, exist_ok=True)
        print("downloading", BASE + name, flush=True)
        with urllib.request.urlopen(BASE + name) as r:
            zipfile.ZipFile(io.BytesIO(r.read())).extractall(d)
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith((".pt", ".pth")):
                return os.path.join(base, f)
    raise FileNotFoundError(f"no .pt file inside {name}")


def val_items(dataset: str, root: str):
    """[(image path, label path)] of the validation split, the reference's file lists."""
    from glob import glob
    if dataset == "cs":
        imgs = sorted(glob(os.path.join(root, "cityscapes", "leftImg8bit", "val", "*", "*.png")))
        return [(p, p.replace("leftImg8bit", "gtFine", 1).replace("_leftImg8bit.png", "_gtFine_labelIds.png")) for p in imgs]
    if dataset == "cv":
        imgs = sorted(glob(os.path.join(root, "camvid", "val", "*.png")))
        return [(p, p.replace(os.sep + "val" + os.sep, os.sep + "valannot" + os.sep)) for p in imgs]
    voc = os.path.join(root, "VOCdevkit", "VOC2012")
    ids = open(os.path.join(voc, "ImageSets", "Segmentation", "val.txt")).read().split()
    return [(os.path.join(voc, "JPEGImages", i + ".jpg"), os.path.join(voc, "SegmentationClass", i + ".png")) for i in ids]


def load_pair(dataset: str, p_img: str, p_lab: str):
    import torch
    from PIL import Image
    x = Image.open(p_img).convert("RGB")
    y = Image.open(p_lab)
    if dataset == "cs":                    # the reference evaluates Cityscapes at quarter resolution (datasets/cityscapes.py: 256 x 512)
        x = x.resize((512, 256), Image.BILINEAR)
        y = y.resize((512, 256), Image.NEAREST)
    ya = np.array(y, dtype=np.uint8)
    if dataset == "cs":
        ya = CS_MAP[ya]
    xa = (np.asarray(x, np.float32) / 255.0 - np.array(MEAN, np.float32)) / np.array(STD, np.float32)
    return torch.from_numpy(xa).permute(2, 0, 1).contiguous(), torch.from_numpy(ya.astype(np.int64))


def evaluate(key: str, ckpt: str, root: str) -> float:
    import torch
    import torch.nn.functional as F
    from pixelpick_amd.utils.metrics import RunningScore
    from pixelpick_amd.utils.utils import get_model
    _, dataset, net, C, _ignore, _ = MODELS[key]
    args = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=net, weight_type="random", n_layers=50,
                     use_softmax=True, use_dilated_resnet=True, width_multiplier=1.0, dataset_name=dataset)
    os.environ.setdefault("PIXELPICK_MNV2_WEIGHTS", "random")      # every weight comes from the checkpoint
    model = get_model(args)
    sd = torch.load(ckpt, map_location="cpu")
    model.load_state_dict(sd["model"] if "model" in sd else sd)    # model.py:208-213 writes {"model": state_dict}
    model = model.cuda().eval()
    score = RunningScore(C)
    with torch.no_grad():
        for p_img, p_lab in val_items(dataset, root):
            x, y = load_pair(dataset, p_img, p_lab)
            h, w = y.shape
            ph, pw = (-h) % 32, (-w) % 32                          # VOC: reflect-pad to the stride, crop back (model.py:181-189, query.py:174)
            xb = x[None].cuda()
            if ph or pw:
                xb = F.pad(xb, (0, pw, 0, ph), mode="reflect")
            pred = model(xb)["pred"][:, :, :h, :w]
            score.update_from_logits(y[None].cuda(), pred.contiguous())      # labels outside [0, C) - the void id - are not counted (utils/metrics.py:168-177)
    return 100.0 * float(score.get_scores()[0]["Mean IoU"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True, help="directory holding the datasets; checkpoints go to <root>/pixelpick_ckpt")
    ap.add_argument("--only", choices=sorted(MODELS), action="append")
    ap.add_argument("--download-only", action="store_true")
    a = ap.parse_args()
    bad = 0
    for key in a.only or sorted(MODELS):
        ckpt = fetch(key, os.path.join(a.root, "pixelpick_ckpt"))
        if a.download_only:
            print(key, "->", ckpt)
            continue
        miou = evaluate(key, ckpt, a.root)
        ref = MODELS[key][5]
        ok = abs(miou - ref) <= 0.5
        bad += not ok
        print(f"{key:10s} mIoU {miou:5.1f}   README {ref:5.1f}   {'ok' if ok else 'DIFFERS'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
