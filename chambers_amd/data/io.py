"""File matching and image decoding of the input pipeline (reference: chambers/data/io.py).

Host side of SURVEY §8f rank 3.  The reference does this with TF string / io ops inside tf.data; here it is plain Python (glob +
PIL) producing numpy uint8 HWC arrays, which `chambers_amd.data.device.DeviceBatcher` packs and hands to the GPU.  The URL helpers
of the reference (io.py:85-113) need network access and are not provided.
"""
import glob
import os

import numpy as np

VALID_IMAGE_EXTENTIONS = ["jpg", "jpeg", "png", "bmp", "gif", "JPG", "JPEG", "PNG", "BMP", "GIF"]   # (sic) io.py:7-18
_MODES = {1: "L", 3: "RGB", 4: "RGBA"}


def validate_dir_path(dir_path):
    """Directory path with a trailing separator (io.py:21-25)."""
    dir_path = os.fspath(dir_path)
    return dir_path if dir_path.endswith("/") else dir_path + "/"


def match_nested_set(path):
    """Sub-directories of `path`, each with a trailing separator, in glob order (io.py:28-29) — callers sort."""
    return glob.glob(os.path.join(path, "*/"))


def match_img_files(dir_path):
    """All files of `dir_path` with a valid image extension, sorted (tf.io.matching_files sorts the union of its patterns;
    io.py:32-52).  A missing directory matches nothing."""
    d = validate_dir_path(dir_path)
    found = set()
    for ext in VALID_IMAGE_EXTENTIONS:
        found.update(glob.glob(glob.escape(d) + "*." + ext))
    return sorted(found)


def match_img_files_triplet(dir_path):
    """(anchor, positive, negative) file lists of a triplet directory (io.py:55-67)."""
    d = validate_dir_path(dir_path)
    return tuple(match_img_files(d + sub) for sub in ("anchor", "positive", "negative"))


def read_and_decode_image(file, channels=3):
    """Decode a .png / .jpeg / .bmp / .gif file to a uint8 [H, W, channels] array; animations yield their first frame
    (tf.image.decode_image(..., expand_animations=False), io.py:70-82)."""
    if channels not in _MODES:
        raise ValueError("channels must be 1, 3 or 4")
    from PIL import Image
    with Image.open(file) as im:
        im.seek(0)
        arr = np.asarray(im.convert(_MODES[channels]), dtype=np.uint8)
    return arr.reshape(arr.shape[0], arr.shape[1], channels)
