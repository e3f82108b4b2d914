#!/usr/bin/env python3
"""Golden vectors for the sequence loader: builds the seeded synthetic EventScape-shaped tree of
recipe.make_dataset_dir, runs the REFERENCE's dataset classes on it (imported from /root/reference, with skimage.io.imread
served by PIL and cv2 stubbed — both absent from this image) and stores what they return.
    python tests/golden/make_golden_dataset.py        (build container only; writes tests/golden/dataset.npz)"""
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from recipe import DATASET_CASES, FOLDERS, flatten_sequence, make_dataset_dir  # noqa: E402
REF = "/root/reference/RAM_Net"


def import_reference():
    from PIL import Image
    sys.path.insert(0, REF)
    sk, io = types.ModuleType("skimage"), types.ModuleType("skimage.io")
    io.imread = lambda path, as_gray=False: np.asarray(Image.open(path))
    sk.io = io
    sys.modules.update({"skimage": sk, "skimage.io": io, "cv2": types.ModuleType("cv2")})
    if not hasattr(np, "alltrue"):
        np.alltrue = np.all
    import data_loader.dataset as D
    import utils.data_augmentation as A
    import bisect
    from torch.utils.data import ConcatDataset
    # train.py cannot be imported without its trainer stack; its 12-line ConcatDatasetCustom.__getitem__ is exercised
    # through the same bisect arithmetic here (dataset index = bisect_right(cumulative_sizes, idx))
    return D, A, bisect, ConcatDataset


def main():
    D, A, bisect, ConcatDataset = import_reference()
    root = make_dataset_dir(tempfile.mkdtemp())
    names = sorted(os.listdir(root))
    out = {}

    def transform(spec):
        if spec is None:
            return None
        return A.CenterCrop(spec[1]) if spec[0] == "center" else A.Compose([A.RandomRotationFlip(0.0, 0.5, 0.0), A.RandomCrop(spec[1])])

    for name, seq_i, idx, seed, spec, kw in DATASET_CASES:
        ds = D.SequenceSynchronizedFramesEventsDataset(os.path.join(root, names[seq_i]), transform=transform(spec), **FOLDERS, **kw)
        random.seed(seed)
        np.random.seed(seed)
        flatten_sequence(name, ds[idx], out)
        out[name + "/len"] = np.asarray([len(ds), ds.dataset.length, ds.event_dataset.first_valid_idx, ds.event_dataset.last_valid_idx])
    # concatenation over the sub-directories, with the dataset index (train.py:23-34, 70-73)
    kw = dict(sequence_length=2, step_size=1, every_x_rgb_frame=3, clip_distance=1000.0, reg_factor=5.70378)
    parts = [D.SequenceSynchronizedFramesEventsDataset(os.path.join(root, n), transform=A.CenterCrop(16), **FOLDERS, **kw)
             for n in names]
    cat = ConcatDataset(parts)
    out["concat/sizes"] = np.asarray(cat.cumulative_sizes)
    for idx in (0, len(parts[0]) - 1, len(parts[0]), len(cat) - 1):
        d = bisect.bisect_right(cat.cumulative_sizes, idx)
        random.seed(100 + idx)
        np.random.seed(100 + idx)
        flatten_sequence("concat/%d" % idx, cat[idx], out)
        out["concat/%d/dataset_idx" % idx] = np.asarray([d])
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("dataset.npz: %d arrays, %.1f KB" % (len(out), os.path.getsize(os.path.join(HERE, "dataset.npz")) / 1024))


if __name__ == "__main__":
    main()
