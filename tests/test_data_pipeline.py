"""CPU tier: host side of the input pipeline — chambers_amd.data against the reference's own expectations.

The expected label sequences are the ones the reference's tests assert for its tf.data pipeline (test_units/data/test_dataset.py:
TestImageClassDataset / TestImageTripletDataset / TestInterleaveImageClassTripletDataset, block_bound0 / block_bound1; test_io.py).
tests/golden/sample_data/mnist/train is the reference's MNIST sample (data files of its tests, 10 classes x 3 PNGs of 28x28); its
triplet sample is 9 MB of photographs, so the triplet folders are synthesised here with the same structure — (anchor, positive,
negative) counts (1,3,3), (1,1,2), (1,2,3), (1,3,3), (1,3,3) — which is all the label sequences depend on.  The reference's
`test_random0` sequences come out of TensorFlow's seeded shuffle and cannot be reproduced without it; the shuffled
configurations are checked through their invariants instead."""
import os

import numpy as np
import pytest

from chambers_amd.data import (InterleaveImageClassDataset, InterleaveImageClassTripletDataset, InterleaveImageTripletDataset,
                               SequentialImageDataset, match_img_files, match_nested_set, read_and_decode_image, set_n_parallel)
from chambers_amd.data.dataset import Dataset, _block_iter, _get_input_len, _random_upsample, _shuffle_repeat
from chambers_amd.data.io import match_img_files_triplet

MNIST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_data", "mnist", "train")
TRIPLET_COUNTS = [(1, 3, 3), (1, 1, 2), (1, 2, 3), (1, 3, 3), (1, 3, 3)]
NC, NB = 5, 2


def _labels(td, batched=False):
    if batched:
        return [int(y) for _xb, yb in td.as_numpy_iterator() for y in yb]
    return [int(y) for _x, y in td.as_numpy_iterator()]


@pytest.fixture(scope="module")
def triplet_dirs(tmp_path_factory):
    from PIL import Image
    root = tmp_path_factory.mktemp("triplets")
    rng = np.random.default_rng(3)
    for k, counts in enumerate(TRIPLET_COUNTS):
        for sub, n in zip(("anchor", "positive", "negative"), counts):
            d = root / ("T%02d" % k) / sub
            d.mkdir(parents=True)
            for j in range(n):
                h, w = int(rng.integers(20, 40)), int(rng.integers(20, 40))
                Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(str(d / ("%d.png" % j)))
    return sorted(match_nested_set(str(root)))


def _class_dirs():
    return sorted(match_nested_set(MNIST))


def _kw(**over):
    kw = dict(class_cycle_length=NC, images_per_block=NB, image_channels=3, block_bound=True, sample_block_random=False, shuffle=False,
              reshuffle_iteration=False, buffer_size=1024, seed=None, repeats=None)
    kw.update(over)
    return kw


# ---- io (test_units/data/test_io.py)
def test_match_and_decode():
    assert len(match_nested_set(MNIST)) == 10
    files = match_img_files(os.path.join(MNIST, "1"))
    assert [os.path.basename(f) for f in files] == ["3.png", "6.png", "8.png"]        # sorted, as tf.io.matching_files
    assert match_img_files(os.path.join(MNIST, "no_such_dir")) == []
    f = os.path.join(MNIST, "1", "3.png")
    assert read_and_decode_image(f, channels=1).shape == (28, 28, 1)
    img3 = read_and_decode_image(f, channels=3)
    assert img3.shape == (28, 28, 3) and img3.dtype == np.uint8
    np.testing.assert_array_equal(img3[..., 0], read_and_decode_image(f, channels=1)[..., 0])
    with pytest.raises(ValueError):
        read_and_decode_image(f, channels=2)


def test_extension_filter_and_gif_first_frame(tmp_path):
    from PIL import Image
    a = np.zeros((5, 7, 3), np.uint8)
    Image.fromarray(a).save(str(tmp_path / "b.JPG"))
    Image.fromarray(a).save(str(tmp_path / "a.bmp"))
    (tmp_path / "notes.txt").write_text("x")
    frames = [Image.fromarray(np.full((4, 6, 3), v, np.uint8)) for v in (10, 200)]
    frames[0].save(str(tmp_path / "c.gif"), save_all=True, append_images=frames[1:])
    assert [os.path.basename(f) for f in match_img_files(str(tmp_path))] == ["a.bmp", "b.JPG", "c.gif"]
    g = read_and_decode_image(str(tmp_path / "c.gif"), channels=3)
    assert g.shape == (4, 6, 3) and abs(int(g[0, 0, 0]) - 10) <= 2                     # expand_animations=False: first frame


# ---- helpers (TestGetInputLen, TestBlockIter, TestShuffleRepeat)
def test_get_input_len():
    assert _get_input_len(("a", "b")) == 2
    assert _get_input_len((["a", "b", "c"], [1, 2, 3])) == 3
    with pytest.raises(ValueError):
        _get_input_len(5)


def test_random_upsample_and_block_iter():
    slices = list(range(10))
    up = _random_upsample(slices, 20)
    assert len(up) == 20 and up[:10] == slices and set(up) <= set(slices)
    assert _random_upsample(slices, len(slices)) == slices
    files = match_img_files(os.path.join(MNIST, "0"))
    pairs = [(f, 0) for f in files]
    assert list(_block_iter(files, 0, 2, block_bound=False)) == pairs
    assert list(_block_iter(files, 0, 2, block_bound=True)) == pairs[:2]
    shuffled = list(_block_iter(files, 0, 2, block_bound=False, sample_block_random=True, seed=1))
    assert sorted(shuffled) == sorted(pairs)
    short = list(_block_iter(files[:1], 7, 4, block_bound=True))                       # fewer files than a block: upsampled
    assert len(short) == 4 and all(e == (files[0], 7) for e in short)
    assert all(isinstance(y, np.int64) for _f, y in pairs and list(_block_iter(files, 0, 2)))


def test_shuffle_repeat():
    slices = list(range(10))
    td = Dataset.from_tensor_slices(slices)
    flat = lambda d: [e[0] for e in d.as_numpy_iterator()]
    assert flat(_shuffle_repeat(td, shuffle=False, repeats=None)) == slices
    assert len(flat(_shuffle_repeat(td, shuffle=False, repeats=3))) == 30
    with pytest.raises(ValueError):
        _shuffle_repeat(td, repeats=0)
    once = flat(_shuffle_repeat(td, shuffle=True, buffer_size=10, reshuffle_iteration=False, seed=5))
    assert sorted(once) == slices and once != slices
    twice = flat(_shuffle_repeat(td, shuffle=True, buffer_size=10, reshuffle_iteration=False, seed=None, repeats=2))
    assert twice[:10] == twice[10:]                                                    # same order every pass
    twice = flat(_shuffle_repeat(td, shuffle=True, buffer_size=10, reshuffle_iteration=True, seed=None, repeats=2))
    assert twice[:10] != twice[10:] and sorted(twice[:10]) == sorted(twice[10:]) == slices
    small = flat(td.shuffle(3, seed=0))                                                # a short buffer only moves elements locally
    assert sorted(small) == slices and all(abs(v - k) <= 9 for k, v in enumerate(small)) and small[0] <= 3


def test_interleave_order_is_tf_datas():
    td = Dataset.from_tensor_slices((list("abcde"), [3, 1, 2, 3, 1]))
    out = [x for (x,) in td.interleave(lambda ch, n: [(ch,)] * n, cycle_length=2, block_length=2)]
    # slots [a(3), b(1)]: a a | b | a | slot 1 refilled: c c | slot 0 refilled: d d | (c exhausted) | d | slot 1 refilled: e
    assert out == ["a", "a", "b", "a", "c", "c", "d", "d", "d", "e"]


# ---- class folders (TestImageClassDataset)
def test_class_dataset_block_bound():
    dirs, labels = _class_dirs(), list(range(10))
    td = InterleaveImageClassDataset(class_dirs=dirs, labels=labels, **_kw())
    expect = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9]
    assert _labels(td) == expect and _labels(td.batch(NC * NB), batched=True) == expect
    x, y = next(iter(td))
    assert x.shape == (28, 28, 3) and x.dtype == np.uint8 and isinstance(y, np.int64)
    xb, yb = next(iter(td.batch(NC * NB)))
    assert xb.shape == (10, 28, 28, 3) and yb.dtype == np.int64


def test_class_dataset_unbound_blocks():
    td = InterleaveImageClassDataset(class_dirs=_class_dirs(), labels=list(range(10)), **_kw(block_bound=False))
    expect = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 0, 1, 2, 3, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 5, 6, 7, 8, 9]
    assert _labels(td) == expect and _labels(td.batch(NC * NB), batched=True) == expect


def test_class_dataset_shuffled_invariants():
    kw = _kw(sample_block_random=True, shuffle=True, seed=42)
    td = InterleaveImageClassDataset(class_dirs=_class_dirs(), labels=list(range(10)), **kw)
    got = _labels(td)
    assert len(got) == 20 and sorted(set(got)) == list(range(10))
    assert all(got[k] == got[k + 1] for k in range(0, 20, 2))                          # blocks of two images of one class
    assert _labels(td) == got                                                          # reshuffle_iteration=False: same pass again
    again = InterleaveImageClassDataset(class_dirs=_class_dirs(), labels=list(range(10)), **kw)
    assert _labels(again) == got                                                       # seeded
    other = InterleaveImageClassDataset(class_dirs=_class_dirs(), labels=list(range(10)), **_kw(sample_block_random=True, shuffle=True, seed=7))
    assert _labels(other) != got
    rep = InterleaveImageClassDataset(class_dirs=_class_dirs(), labels=list(range(10)), **_kw(repeats=2))
    assert len(_labels(rep)) == 40


def test_n_parallel_setting():
    try:
        assert InterleaveImageClassDataset(class_dirs=_class_dirs(), labels=list(range(10)), **_kw())._num_parallel_calls == -1
        set_n_parallel(3)
        td = InterleaveImageClassDataset(class_dirs=_class_dirs(), labels=list(range(10)), **_kw())
        assert td._num_parallel_calls == 3
        assert _labels(td) == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9]    # parallel decode keeps the order
    finally:
        set_n_parallel(-1)


# ---- triplet folders (TestImageTripletDataset)
def test_triplet_dataset(triplet_dirs):
    a, p, n = match_img_files_triplet(triplet_dirs[1])
    assert (len(a), len(p), len(n)) == TRIPLET_COUNTS[1]
    labels = list(range(len(triplet_dirs)))
    td = InterleaveImageTripletDataset(class_dirs=triplet_dirs, labels=labels, **_kw())
    expect = [0, -1, 1, -1, 2, -1, 3, -1, 4, -1]
    assert _labels(td) == expect
    td = InterleaveImageTripletDataset(class_dirs=triplet_dirs, labels=labels, **_kw(block_bound=False))
    expect = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 0, 0, -1, -1, 2, -1, 3, 3, 4, 4, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1]
    assert _labels(td) == expect
    td = InterleaveImageTripletDataset(class_dirs=triplet_dirs, labels=labels, **_kw(sample_block_random=True, shuffle=True, seed=42))
    got = _labels(td)
    assert len(got) == 10 and got[1::2] == [-1] * 5 and sorted(got[0::2]) == labels


# ---- class + triplet folders (TestInterleaveImageClassTripletDataset)
def test_class_triplet_dataset(triplet_dirs):
    dirs = _class_dirs() + triplet_dirs
    labels = list(range(len(dirs)))
    td = InterleaveImageClassTripletDataset(class_dirs=dirs, labels=labels, **_kw())
    expect = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, -1, 11, -1, 12, -1, 13, -1, 14, -1]
    assert _labels(td) == expect
    td = InterleaveImageClassTripletDataset(class_dirs=dirs, labels=labels, **_kw(block_bound=False))
    expect = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 0, 1, 2, 3, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 5, 6, 7, 8, 9, 10,
              10, 11, 11, 12, 12, 13, 13, 14, 14, 10, 10, -1, -1, 12, -1, 13, 13, 14, 14, -1, -1, -1, -1,
              -1, -1, -1, -1, -1, -1, -1]
    assert _labels(td) == expect


def test_sequential_dataset():
    td = SequentialImageDataset(class_dirs=_class_dirs(), labels=list(range(10)))
    assert _labels(td) == [c for c in range(10) for _ in range(3)]
    td = SequentialImageDataset(class_dirs=_class_dirs(), labels=list(range(10)), shuffle=True, seed=1, repeats=2)
    got = _labels(td)
    assert len(got) == 60 and all(got[k] == got[k + 1] == got[k + 2] for k in range(0, 60, 3))
