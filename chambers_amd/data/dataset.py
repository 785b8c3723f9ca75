"""Class-interleaved image datasets (reference: chambers/data/dataset.py).

The reference builds these from tf.data (from_tensor_slices -> shuffle/repeat -> interleave over per-class block iterators ->
map(read_and_decode_image)).  There is no TensorFlow here: `Dataset` below is a small re-iterable pipeline with the handful of
tf.data operations those builders use, with the same element order for every deterministic configuration (the reference's own
expected label sequences are the tests, tests/test_data_pipeline.py).  Seeded shuffles use numpy's PCG64, so a seeded order is
reproducible here but is not TensorFlow's order.

Elements are tuples; `map` / `interleave` / `flat_map` call their function with the tuple unpacked, as tf.data does.
"""
import itertools
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .io import match_img_files, match_img_files_triplet, read_and_decode_image

_CONFIG = {"N_PARALLEL": -1}          # -1 = tf.data.AUTOTUNE in the reference (dataset.py:12)
_ENTROPY = itertools.count()


def set_n_parallel(n):
    """Worker count of the decode `map` of datasets built afterwards (dataset.py:15-16)."""
    _CONFIG["N_PARALLEL"] = n


def _workers(n):
    if n is None:
        return 0
    if n == -1:
        return min(32, os.cpu_count() or 1)
    return max(0, int(n))


def _as_tuple(e):
    return e if isinstance(e, tuple) else (e,)


class Dataset:
    """Re-iterable element pipeline: every `iter()` starts a fresh pass (as a tf.data.Dataset does)."""

    def __init__(self, make_iter, num_parallel_calls=None):
        self._make_iter = make_iter
        self._num_parallel_calls = num_parallel_calls

    def __iter__(self):
        return self._make_iter()

    def as_numpy_iterator(self):
        return iter(self)

    # ---- sources
    @staticmethod
    def from_tensor_slices(inputs):
        if isinstance(inputs, tuple):
            cols = [list(c) for c in inputs]
            if len({len(c) for c in cols}) > 1:
                raise ValueError("all components must have the same length")
            rows = list(zip(*cols))
        else:
            rows = [(v,) for v in inputs]
        return Dataset(lambda: iter(rows))

    # ---- element-wise
    def map(self, fn, num_parallel_calls=None):
        nw = _workers(num_parallel_calls)

        def gen():
            if nw <= 1:
                for e in self:
                    yield _as_tuple(fn(*e))
                return
            with ThreadPoolExecutor(max_workers=nw) as pool:      # ordered, at most 2 * nw elements in flight
                window = []
                for e in self:
                    window.append(pool.submit(fn, *e))
                    if len(window) >= 2 * nw:
                        yield _as_tuple(window.pop(0).result())
                for f in window:
                    yield _as_tuple(f.result())

        return Dataset(gen, num_parallel_calls)

    def flat_map(self, fn):
        def gen():
            for e in self:
                for sub in fn(*e):
                    yield sub
        return Dataset(gen)

    def interleave(self, fn, cycle_length, block_length=1, num_parallel_calls=None):
        """tf.data interleave, deterministic order: `cycle_length` input elements are open at once and visited round-robin,
        `block_length` consecutive outputs per visit; an exhausted slot passes the turn on and is refilled with the next input
        element when it is visited again."""
        if cycle_length < 1 or block_length < 1:
            raise ValueError("cycle_length and block_length must be >= 1")

        def gen():
            inputs = iter(self)
            slots = [None] * cycle_length
            more_inputs = True
            c = 0
            while True:
                if slots[c] is None and more_inputs:
                    try:
                        slots[c] = iter(fn(*next(inputs)))
                    except StopIteration:
                        more_inputs = False
                if slots[c] is not None:
                    for _ in range(block_length):
                        try:
                            yield next(slots[c])
                        except StopIteration:
                            slots[c] = None
                            break
                elif not more_inputs and all(s is None for s in slots):
                    return
                c = (c + 1) % cycle_length

        return Dataset(gen, num_parallel_calls)

    # ---- order / length
    def shuffle(self, buffer_size, seed=None, reshuffle_each_iteration=True):
        """Buffered uniform shuffle (a buffer as long as the data is a full permutation)."""
        if buffer_size is None or buffer_size < 1:
            raise ValueError("buffer_size must be >= 1")
        base = seed if seed is not None else (int.from_bytes(os.urandom(4), "little") + next(_ENTROPY))
        epoch = itertools.count()

        def gen():
            rng = np.random.Generator(np.random.PCG64([base, next(epoch) if reshuffle_each_iteration else 0]))
            buf = []
            for e in self:
                buf.append(e)
                if len(buf) > buffer_size:
                    k = int(rng.integers(len(buf)))
                    buf[k], buf[-1] = buf[-1], buf[k]
                    yield buf.pop()
            while buf:
                k = int(rng.integers(len(buf)))
                buf[k], buf[-1] = buf[-1], buf[k]
                yield buf.pop()

        return Dataset(gen)

    def repeat(self, count=None):
        def gen():
            n = 0
            while count is None or count == -1 or n < count:
                empty = True
                for e in self:
                    empty = False
                    yield e
                if empty:
                    return
                n += 1
        return Dataset(gen)

    def take(self, count):
        return Dataset(lambda: itertools.islice(iter(self), int(count)))

    def concatenate(self, other):
        return Dataset(lambda: itertools.chain(iter(self), iter(other)))

    def batch(self, batch_size, drop_remainder=False):
        """Tuples of stacked components; a component whose arrays differ in shape (undecoded sizes) stays a list."""
        def stack(vals):
            arrs = [np.asarray(v) for v in vals]
            if len({a.shape for a in arrs}) == 1:
                return np.stack(arrs)
            return arrs

        def gen():
            it = iter(self)
            while True:
                chunk = list(itertools.islice(it, batch_size))
                if not chunk or (drop_remainder and len(chunk) < batch_size):
                    return
                yield tuple(stack(col) for col in zip(*chunk))

        return Dataset(gen)


# ---------------------------------------------------------------------------------------------
def _shuffle_repeat(dataset, shuffle=False, buffer_size=None, reshuffle_iteration=True, seed=None, repeats=None):
    """dataset.py:19-40: optional shuffle, then `repeats` passes (-1 = forever, None = one pass)."""
    if shuffle:
        dataset = dataset.shuffle(buffer_size=buffer_size, seed=seed, reshuffle_each_iteration=reshuffle_iteration)
    if repeats is not None:
        if not (repeats == -1 or repeats > 0):
            raise ValueError("'repeats' must be greater than zero or equal to -1.")
        dataset = dataset.repeat(repeats)
    return dataset


def _get_input_len(inputs):
    """Length of a 1-D input, or of the first component of a tuple of them (dataset.py:43-52)."""
    nd = np.ndim(inputs)
    if nd == 0:
        raise ValueError("Input with 0 dimensions has no length.")
    return len(inputs) if nd == 1 else len(inputs[0])


def _sequential_dataset(inputs, shuffle=False, reshuffle_iteration=True, buffer_size=None, seed=None, repeats=None):
    """dataset.py:55-75."""
    n = _get_input_len(inputs)
    td = Dataset.from_tensor_slices(tuple(inputs) if np.ndim(inputs) > 1 else list(inputs))
    return _shuffle_repeat(td, shuffle=shuffle, buffer_size=buffer_size or n,
                           reshuffle_iteration=reshuffle_iteration, seed=seed, repeats=repeats)


def _random_upsample(x, n, seed=None):
    """`x` padded to length n with uniformly drawn members of x (dataset.py:78-87)."""
    x = list(x)
    if not x:
        raise ValueError("cannot upsample an empty block")
    if n <= len(x):
        return x
    rng = np.random.Generator(np.random.PCG64(seed))
    return x + [x[int(k)] for k in rng.integers(0, len(x), size=n - len(x))]


def _block_iter(block_tensor, label, block_length, block_bound=True, sample_block_random=False, seed=None):
    """One class' contribution to the interleave (dataset.py:90-121): its files with the class label; short classes are upsampled
    to a full block, `sample_block_random` shuffles the class, `block_bound` keeps one block of it."""
    files = list(block_tensor)
    block_length = int(block_length)
    if len(files) < block_length:
        files = _random_upsample(files, block_length)
    td = Dataset.from_tensor_slices((files, [np.int64(label)] * len(files)))
    if sample_block_random and files:
        td = td.shuffle(len(files), seed=seed)
    if block_bound:
        td = td.take(block_length)
    return td


def _block_iter_triplet(triplets, label, block_length, block_bound=True, sample_block_random=False, seed=None):
    """dataset.py:124-157: anchors + positives carry the label in the first floor(block/2) slots, negatives carry -1 in the rest."""
    anchor, positive, negative = triplets
    kw = dict(block_bound=block_bound, sample_block_random=sample_block_random, seed=seed)
    pos = _block_iter(list(anchor) + list(positive), label, block_length // 2, **kw)
    neg = _block_iter(negative, -1, block_length - block_length // 2, **kw)
    return pos.concatenate(neg)


def _class_block(kind, block_length, block_bound, sample_block_random, seed):
    """fn(dir, label) -> block Dataset for the three directory layouts (dataset.py:160-247): 'class' = image files in the
    directory, 'triplet' = anchor/positive/negative sub-directories, 'either' = triplet layout when the directory itself holds no
    image."""
    kw = dict(block_length=block_length, block_bound=block_bound, sample_block_random=sample_block_random, seed=seed)

    def fn(input_dir, label):
        if kind != "triplet":
            files = match_img_files(input_dir)
            if kind == "class" or files:
                return _block_iter(files, label, **kw)
        return _block_iter_triplet(match_img_files_triplet(input_dir), label, **kw)

    return fn


def _interleave_dataset(inputs, interleave_fn, cycle_length, block_length, shuffle=False, reshuffle_iteration=True, buffer_size=None,
                        seed=None, repeats=None):
    """dataset.py:250-271."""
    td = _sequential_dataset(inputs, shuffle=shuffle, reshuffle_iteration=reshuffle_iteration, buffer_size=buffer_size, seed=seed,
                             repeats=repeats)
    return td.interleave(interleave_fn, cycle_length=cycle_length, block_length=block_length, num_parallel_calls=_CONFIG["N_PARALLEL"])


def _decoded(td, image_channels):
    return td.map(lambda f, y: (read_and_decode_image(f, channels=image_channels), y), num_parallel_calls=_CONFIG["N_PARALLEL"])


def _interleave_images(kind, class_dirs, labels, class_cycle_length, images_per_block, image_channels, block_bound, sample_block_random,
                       shuffle, reshuffle_iteration, buffer_size, seed, repeats):
    if images_per_block is None or images_per_block == -1:
        images_per_block = 1
    td = _interleave_dataset((list(class_dirs), list(labels)), _class_block(kind, images_per_block, block_bound, sample_block_random, seed),
                             cycle_length=class_cycle_length, block_length=images_per_block, shuffle=shuffle,
                             reshuffle_iteration=reshuffle_iteration, buffer_size=buffer_size, seed=seed, repeats=repeats)
    return _decoded(td, image_channels)


def InterleaveImageClassDataset(class_dirs, labels, class_cycle_length, images_per_block, image_channels=3, block_bound=True,
                                sample_block_random=False, shuffle=False, reshuffle_iteration=True, buffer_size=None, seed=None,
                                repeats=None):
    """(image uint8 [H,W,C], label) elements, interleaving `class_cycle_length` class folders `images_per_block` images at a time
    (dataset.py:264-315)."""
    return _interleave_images("class", class_dirs, labels, class_cycle_length, images_per_block, image_channels, block_bound,
                              sample_block_random, shuffle, reshuffle_iteration, buffer_size, seed, repeats)


def InterleaveImageTripletDataset(class_dirs, labels, class_cycle_length, images_per_block, image_channels=3, block_bound=True,
                                  sample_block_random=False, shuffle=False, reshuffle_iteration=True, buffer_size=None, seed=None,
                                  repeats=None):
    """The same over triplet folders (anchor / positive / negative sub-folders; negatives are labelled -1; dataset.py:318-363)."""
    return _interleave_images("triplet", class_dirs, labels, class_cycle_length, images_per_block, image_channels, block_bound,
                              sample_block_random, shuffle, reshuffle_iteration, buffer_size, seed, repeats)


def InterleaveImageClassTripletDataset(class_dirs, labels, class_cycle_length, images_per_block, image_channels=3, block_bound=True,
                                       sample_block_random=False, shuffle=False, reshuffle_iteration=True, buffer_size=None, seed=None,
                                       repeats=None):
    """Class folders and triplet folders mixed: a folder without images of its own is read as a triplet folder
    (dataset.py:366-411)."""
    return _interleave_images("either", class_dirs, labels, class_cycle_length, images_per_block, image_channels, block_bound,
                              sample_block_random, shuffle, reshuffle_iteration, buffer_size, seed, repeats)


def SequentialImageDataset(class_dirs, labels, image_channels=3, shuffle=False, reshuffle_iteration=True, buffer_size=None, seed=None,
                           repeats=None):
    """Every image of every folder, folder after folder (dataset.py:414-438)."""
    td = _sequential_dataset((list(class_dirs), list(labels)), shuffle=shuffle, reshuffle_iteration=reshuffle_iteration,
                             buffer_size=buffer_size, seed=seed, repeats=repeats)
    td = td.flat_map(lambda d, y: [(f, np.int64(y)) for f in match_img_files(d)])
    return _decoded(td, image_channels)
