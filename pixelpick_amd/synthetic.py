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
        return {'x': self.xs[i], 'y': self.ys[i], 'queries': torch.from_numpy(self.queries[i]), 'p_img': self.names[i]}
