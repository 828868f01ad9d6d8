"""In-memory dataset with the reference's dataset surface (datasets/base_dataset.py:14-45,182-188), filled with
synthetic images: blobs of class colour + noise, so that a few training steps are learnable.  Used by the driver
tests and by `tools/run_al_synthetic.py`; the reference's file-based datasets are out of scope."""
import numpy as np
import torch

from .query import QuerySelector


class SyntheticDataset(torch.utils.data.Dataset):
    def __init__(self, n_images, height, width, n_classes, ignore_index, n_init_pixels=0, seed=0, void_fraction=0.03):
        rng = np.random.RandomState(seed)
        self.n_classes, self.ignore_index = n_classes, ignore_index
        self.xs, self.ys, self.names = [], [], []
        proto = rng.randn(n_classes, 3).astype(np.float32) * 1.5
        for i in range(n_images):
            # coarse random label map upsampled x8 -> contiguous regions
            coarse = rng.randint(0, n_classes, size=((height + 7) // 8, (width + 7) // 8))
            y = np.kron(coarse, np.ones((8, 8), dtype=np.int64))[:height, :width]
            x = proto[y].transpose(2, 0, 1) + rng.randn(3, height, width).astype(np.float32) * 0.3
            y = y.copy()
            y[rng.rand(height, width) < void_fraction] = ignore_index
            self.xs.append(torch.from_numpy(x.astype(np.float32)))
            self.ys.append(torch.from_numpy(y))
            self.names.append(f"synthetic/img_{i:04d}.png")
        self.queries = []
        for i in range(n_images):
            q = np.zeros((height, width), dtype=np.bool_)
            if n_init_pixels > 0:
                q.reshape(-1)[rng.choice(height * width, n_init_pixels, replace=False)] = True
            self.queries.append(q)
        self.n_pixels_total = int(sum(q.sum() for q in self.queries))
        self.labelled_rounds = []
        self.image_sizes = [(height, width)] * n_images
        self.n_getitem = 0

    def label_queries(self, queries, nth_query=None):
        """base_dataset.py:24-45: OR-merge the new masks into self.queries."""
        new = QuerySelector.decode_queries(queries)
        assert len(new) == len(self.queries), f"{len(new)} != {len(self.queries)}"
        for i, q in enumerate(new):
            self.queries[i] = np.logical_or(self.queries[i], q)
        self.n_pixels_total = int(sum(q.sum() for q in self.queries))
        self.labelled_rounds.append(nth_query)

    def __len__(self):
        return len(self.xs)

    def __getitem__(self, i):
        self.n_getitem += 1
        return {'x': self.xs[i], 'y': self.ys[i], 'queries': torch.from_numpy(self.queries[i]), 'p_img': self.names[i]}


class RawSyntheticDataset(torch.utils.data.Dataset):
    """The same surface over RAW data, for the device data path: what the reference's `__getitem__`
    (base_dataset.py:151-196) holds BEFORE augmentation - an RGB uint8 image as PIL decodes it, a uint8 label map, the bool
    query mask - instead of the augmented float tensors its DataLoader workers produce.

    train=True   items are raw: {'x_u8': uint8 [H,W,3], 'y_u8': uint8 [H,W], 'queries': bool [H,W], 'p_img'} - augmentation,
                 to_tensor and normalisation happen on the GPU (pixelpick_amd.augment.DeviceAugmenter, driven by Model).
    train=False  the reference's val / query branch (base_dataset.py:178-181: no augmentation): 'x' = normalize(to_tensor(img)),
                 'y' int64, 'queries' uint8.
    `image_sizes` lists (h, w) of every item without loading it (a sharded acquisition round advances the host RNG streams of
    other ranks' images from it); `n_getitem` counts the items this process really collated."""

    def __init__(self, n_images, height, width, n_classes, ignore_index, mean, std, n_init_pixels=0, seed=0, void_fraction=0.03,
                 train=True, sizes=None):
        rng = np.random.RandomState(seed)
        assert ignore_index < 256 and n_classes <= 256
        self.n_classes, self.ignore_index, self.train = n_classes, ignore_index, train
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.imgs, self.labels, self.names, self.image_sizes = [], [], [], []
        proto = rng.randint(30, 226, size=(n_classes, 3))
        for i in range(n_images):
            h, w = (height, width) if sizes is None else sizes[i % len(sizes)]
            coarse = rng.randint(0, n_classes, size=((h + 7) // 8, (w + 7) // 8))
            y = np.kron(coarse, np.ones((8, 8), dtype=np.int64))[:h, :w]
            x = np.clip(proto[y] + rng.randint(-25, 26, size=(h, w, 3)), 0, 255).astype(np.uint8)
            y = y.astype(np.uint8)
            y[rng.rand(h, w) < void_fraction] = ignore_index
            self.imgs.append(x)
            self.labels.append(y)
            self.names.append(f"synthetic_raw/img_{i:04d}.png")
            self.image_sizes.append((h, w))
        self.queries = []
        for i in range(n_images):
            h, w = self.image_sizes[i]
            q = np.zeros((h, w), dtype=np.bool_)
            if n_init_pixels > 0:
                q.reshape(-1)[rng.choice(h * w, n_init_pixels, replace=False)] = True
            self.queries.append(q)
        self.n_pixels_total = int(sum(q.sum() for q in self.queries))
        self.labelled_rounds = []
        self.list_labelled_queries = None
        self.n_getitem = 0

    def view(self, train: bool):
        """Another dataset object over the same images (the reference builds separate train / query datasets over the same
        files, each with its own `queries`, model.py:27-37)."""
        import copy
        v = copy.copy(self)
        v.train = train
        v.queries = [q.copy() for q in self.queries]
        v.labelled_rounds = []
        v.n_getitem = 0
        return v

    label_queries = SyntheticDataset.label_queries

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, i):
        self.n_getitem += 1
        if self.train:
            return {'x_u8': torch.from_numpy(self.imgs[i]), 'y_u8': torch.from_numpy(self.labels[i]),
                    'queries': torch.from_numpy(self.queries[i]), 'p_img': self.names[i]}
        x = torch.from_numpy(self.imgs[i]).permute(2, 0, 1).float().div(255.0)            # TF.to_tensor
        x = (x - torch.tensor(self.mean).view(3, 1, 1)) / torch.tensor(self.std).view(3, 1, 1)   # TF.normalize
        return {'x': x, 'y': torch.from_numpy(self.labels[i].astype(np.int64)),
                'queries': torch.from_numpy(self.queries[i].astype(np.uint8)), 'p_img': self.names[i]}
