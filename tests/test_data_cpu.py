"""CPU: the sequence loader (rpg_ramnet_amd/data.py) against fixtures produced by the reference's own dataset
classes on the same seeded synthetic tree (tests/golden/make_golden_dataset.py).  Bit-exact: same files, same numpy /
torch operations, same consumption of the random streams."""
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from recipe import DATASET_CASES, FOLDERS, flatten_sequence, make_dataset_dir  # noqa: E402

from rpg_ramnet_amd import data as D  # noqa: E402


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = make_dataset_dir(str(tmp_path_factory.mktemp("eventscape")))
    return root, sorted(os.listdir(root))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "dataset.npz"))


def transform(spec):
    if spec is None:
        return None
    return D.CenterCrop(spec[1]) if spec[0] == "center" else D.Compose([D.RandomRotationFlip(0.0, 0.5, 0.0), D.RandomCrop(spec[1])])


def check(prefix, seq, gold):
    got = {}
    flatten_sequence(prefix, seq, got)
    want = [k for k in gold.files if k.startswith(prefix + "/") and k.split("/")[-2].isdigit() and not k.endswith("dataset_idx")]
    assert sorted(got) == sorted(want), (sorted(got), sorted(want))
    for k in want:
        assert got[k].dtype == gold[k].dtype and got[k].shape == gold[k].shape, k
        assert np.array_equal(got[k], gold[k], equal_nan=True), "%s differs (max %g)" % (k, np.nanmax(np.abs(got[k] - gold[k])))


@pytest.mark.parametrize("case", DATASET_CASES, ids=[c[0] for c in DATASET_CASES])
def test_sequence_dataset_matches_reference(case, tree, gold):
    name, seq_i, idx, seed, spec, kw = case
    root, names = tree
    ds = D.SequenceSynchronizedFramesEventsDataset(os.path.join(root, names[seq_i]), transform=transform(spec), **FOLDERS, **kw)
    assert [len(ds), ds.dataset.length, ds.event_dataset.first_valid_idx, ds.event_dataset.last_valid_idx] == list(gold[name + "/len"])
    random.seed(seed)
    np.random.seed(seed)
    check(name, ds[idx], gold)


def test_concatenated_subfolders_with_dataset_index(tree, gold):
    root, names = tree
    kw = dict(sequence_length=2, step_size=1, every_x_rgb_frame=3, clip_distance=1000.0, reg_factor=5.70378)
    parts = [D.SequenceSynchronizedFramesEventsDataset(os.path.join(root, n), transform=D.CenterCrop(16), **FOLDERS, **kw) for n in names]
    cat = D.ConcatDatasetCustom(parts)
    assert list(cat.cumulative_sizes) == list(gold["concat/sizes"])
    for idx in (0, len(parts[0]) - 1, len(parts[0]), len(cat) - 1):
        random.seed(100 + idx)
        np.random.seed(100 + idx)
        seq, d = cat[idx]
        assert d == int(gold["concat/%d/dataset_idx" % idx][0])
        check("concat/%d" % idx, seq, gold)
    # the factory of train.py:37-75 (os.listdir order, optional dataset index)
    both = D.concatenate_subfolders(root, "SequenceSynchronizedFramesEventsDataset", sequence_length=2, transform=D.CenterCrop(16),
                                    dataset_idx_flag=True, **FOLDERS, **{k: v for k, v in kw.items() if k != "sequence_length"})
    assert len(both) == len(cat) and isinstance(both[0], tuple)


def test_loader_edge_cases(tree):
    root, names = tree
    base = os.path.join(root, names[0])
    kw = dict(step_size=1, every_x_rgb_frame=3, clip_distance=1000.0)
    assert len(D.SequenceSynchronizedFramesEventsDataset(base, sequence_length=5, **FOLDERS, **kw)) == 0      # L*K >= number of grids
    ds = D.SequenceSynchronizedFramesEventsDataset(base, sequence_length=2, proba_pause_when_running=1.0, **FOLDERS, **kw)
    with pytest.raises(KeyError):       # the reference's pause augmentation indexes item['events'] and fails the same way
        ds[0]
    with pytest.raises(AssertionError):
        D.SequenceSynchronizedFramesEventsDataset(base, sequence_length=0, **FOLDERS, **kw)
    g = D.VoxelGridDataset(base, "events/voxels")[3]["events"]          # the all-zero grid stays untouched by the normalisation
    assert float(g.abs().max()) == 0.0
    t = D.SynchronizedFramesEventsDataset(base, clip_distance=1000.0, every_x_rgb_frame=1, **FOLDERS).__getitem__(2, seed=1)
    d = t["depth_image"]
    assert d.shape[0] == 1 and float(d[~torch.isnan(d)].min()) >= 0.0 and float(d[~torch.isnan(d)].max()) <= 1.0 and bool(torch.isnan(d).any())


def test_sharded_sequence_sampler_partitions_the_index_list():
    """SURVEY 8e partitioning on the real loader: ranks get disjoint sequences r, r + world, ... of one shared shuffled order,
    the same count on every rank, a different order per epoch."""
    from rpg_ramnet_amd.data import ShardedSequenceSampler
    ds = list(range(37))
    world = 4
    per_epoch = []
    for epoch in (0, 1):
        got = []
        for r in range(world):
            sm = ShardedSequenceSampler(ds, r, world, batch_size=2)
            sm.set_epoch(epoch)
            idx = list(sm)
            assert len(idx) == len(sm) == 8                      # 37 // (4 * 2) batches of 2
            got.append(idx)
        flat = [i for g in got for i in g]
        assert len(set(flat)) == len(flat) and set(flat) <= set(ds)
        per_epoch.append(got)
    assert per_epoch[0] != per_epoch[1]
    plain = [list(ShardedSequenceSampler(ds, r, world, shuffle=False)) for r in range(world)]
    assert plain[1][:3] == [1, 5, 9] and all(len(p) == 9 for p in plain)


def test_device_prefetcher_passes_sequences_through_on_cpu():
    """data.DevicePrefetcher on a CPU device: same structure, same values, every element of the loader exactly once, in order."""
    import torch
    from rpg_ramnet_amd.data import DevicePrefetcher
    seqs = [[{"image": torch.full((1, 1, 2, 2), float(10 * s + l)), "events0": torch.zeros(1, 5, 2, 2), "meta": "k%d" % s} for l in range(3)]
            for s in range(4)]
    out = list(DevicePrefetcher(seqs, "cpu"))
    assert len(out) == 4 and all(len(o) == 3 for o in out)
    for s, o in enumerate(out):
        for l, item in enumerate(o):
            assert item["meta"] == "k%d" % s and float(item["image"].mean()) == 10 * s + l and item["events0"].shape == (1, 5, 2, 2)
    assert list(DevicePrefetcher([], "cpu")) == []
