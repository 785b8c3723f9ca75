"""Pins the oracle: reference-derived known-answer vectors (ImageNetNormalization, the only ones the
reference's tests hold for this path) + builder-derived KATs for the TFA-backed ops on tiny images."""
import numpy as np
import pytest

from oracle import augment_ref as A

# test_units/augmentations/test_image_augmentations.py:5-15
IMG = np.array([[139, 186, 208, 200], [175, 201, 198, 200], [166, 191, 193, 195], [124, 155, 172, 151]], dtype=np.uint8)
IMG = np.stack([IMG, IMG, IMG], axis=-1)[None]

# :21-64 (reference-derived goldens, exact equality)
TARGETS = {
    "caffe": [[35.060997, 82.061, 104.061, 96.061], [71.061, 97.061, 94.061, 96.061], [62.060997, 87.061, 89.061, 91.061],
              [20.060997, 51.060997, 68.061, 47.060997]],
    "tf": [[0.0901961327, 0.458823562, 0.631372571, 0.568627477], [0.372549057, 0.576470613, 0.552941203, 0.568627477],
           [0.301960826, 0.498039246, 0.513725519, 0.529411793], [-0.0274509788, 0.215686321, 0.349019647, 0.184313774]],
    "torch": [[0.262436897, 1.06730032, 1.44404483, 1.30704677], [0.878928, 1.32417154, 1.27279735, 1.30704677],
              [0.724805236, 1.15292406, 1.1871736, 1.22142303], [0.00556548592, 0.536432922, 0.827553749, 0.467933923]],
}


@pytest.mark.parametrize("mode", ["caffe", "tf", "torch"])
def test_imagenet_normalization_reference_kat(mode):
    out = A.imagenet_normalize(IMG, mode)[0, ..., 0]
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, np.array(TARGETS[mode], dtype=np.float32))


def test_unknown_mode_raises():
    with pytest.raises(ValueError):
        A.imagenet_normalize(IMG, "nope")


# ---- builder-derived KATs (hand-computed from the restated upstream algorithms) ----------------
def test_invert_posterize_solarize_kat():
    x = np.array([[[[0, 1, 127]], [[128, 200, 255]]]], dtype=np.uint8)
    np.testing.assert_array_equal(A.invert(x).ravel(), [255, 254, 128, 127, 55, 0])
    np.testing.assert_array_equal(A.posterize(x, 3).ravel(), [0, 0, 96, 128, 192, 224])
    np.testing.assert_array_equal(A.posterize(x, 0).ravel(), [0, 0, 0, 128, 128, 128])  # shift clamped to 7
    np.testing.assert_array_equal(A.solarize(x, 128).ravel(), [0, 1, 127, 127, 55, 0])
    np.testing.assert_array_equal(A.solarize(x, 256).ravel(), x.ravel())
    np.testing.assert_array_equal(A.solarize_add(x, 99, 128).ravel(), [99, 100, 226, 128, 200, 255])


def test_blend_and_brightness_kat():
    x = np.array([[[[10, 100, 200]]]], dtype=np.uint8)
    np.testing.assert_array_equal(A.brightness(x, 1.72).ravel(), [17, 172, 255])   # 17.2, 172, 344 -> clip
    np.testing.assert_array_equal(A.brightness(x, 0.5).ravel(), [5, 50, 100])
    np.testing.assert_array_equal(A.brightness(x, 1.0).ravel(), x.ravel())
    np.testing.assert_array_equal(A.brightness(x, 0.0).ravel(), [0, 0, 0])


def test_contrast_constant_quirk():
    assert A.contrast_constant(224 * 224) == 196      # one 224^2 image
    assert A.contrast_constant(2 * 224 * 224) == 255  # any batch >= 2 saturates
    assert A.contrast_constant(16) == 0


def test_grayscale_kat():
    x = np.array([[[[255, 255, 255]], [[255, 0, 0]], [[0, 255, 0]], [[0, 0, 255]]]], dtype=np.uint8)
    # 0.2989+0.5870+0.1140 = 0.9999 -> 255.47 -> 255 ; 0.2989*255.5 = 76.4 ; 0.587*255.5 = 149.97 ; 0.114*255.5 = 29.1
    np.testing.assert_array_equal(A.rgb_to_grayscale(x).ravel(), [255, 76, 149, 29])


def test_equalize_kat():
    # 4 pixels, one channel: values 0,0,128,255 -> hist nonzero [2,1,1]; step = (4 - 1)//255 = 0 -> identity
    x = np.array([0, 0, 128, 255], dtype=np.uint8).reshape(1, 2, 2, 1)
    np.testing.assert_array_equal(A.equalize(x), x)
    # 16x32 ramp: 512 pixels, values 0..255 twice: step = (512-2)//255 = 2; lut[v] = (2v + 1)//2 = v
    y = np.tile(np.arange(256, dtype=np.uint8), 2).reshape(1, 16, 32, 1)
    np.testing.assert_array_equal(A.equalize(y), y)
    # constant image: step = 0 -> identity
    z = np.full((1, 4, 4, 1), 7, dtype=np.uint8)
    np.testing.assert_array_equal(A.equalize(z), z)
    # skewed: 12 zeros + 4 of value 200 over 16 px: step = (16-4)//255 = 0 -> identity
    w = np.array([0] * 12 + [200] * 4, dtype=np.uint8).reshape(1, 4, 4, 1)
    np.testing.assert_array_equal(A.equalize(w), w)
    # 1024 px: 512 x value 10, 256 x 20, 256 x 30: step = (1024-256)//255 = 3;
    # lut[10] = (0+1)//3 = 0, lut[20] = (512+1)//3 = 171, lut[30] = (768+1)//3 = 256 -> 255
    v = np.array([10] * 512 + [20] * 256 + [30] * 256, dtype=np.uint8).reshape(1, 32, 32, 1)
    out = A.equalize(v).ravel()
    assert set(out[:512]) == {0} and set(out[512:768]) == {171} and set(out[768:]) == {255}


def test_translate_and_cutout_kat():
    x = np.arange(16, dtype=np.uint8).reshape(1, 4, 4, 1)
    t = A.translate_x_transform(1.0, negate=False)      # in_x = out_x + 1
    out = A.projective_transform(x, t, fill_value=128)[0, :, :, 0]
    np.testing.assert_array_equal(out[0], [1, 2, 3, 128])
    t = A.translate_y_transform(2.0, negate=True)       # in_y = out_y - 2
    out = A.projective_transform(x, t, fill_value=128)[0, :, :, 0]
    np.testing.assert_array_equal(out[:, 0], [128, 128, 0, 4])
    c = A.cutout(x, 2, [[0, 3]], 9)[0, :, :, 0]          # rows [-1,1) cols [2,4)
    np.testing.assert_array_equal(c[0], [0, 1, 9, 9])
    np.testing.assert_array_equal(c[1], [4, 5, 6, 7])
    with pytest.raises(ValueError):
        A.cutout(x, 3, [[0, 0]], 0)


def test_round_half_away():
    v = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 0.49999997], dtype=np.float32)
    np.testing.assert_array_equal(A._round_half_away(v), [1, 2, 3, -1, -2, 0])


def test_sharpness_kat():
    x = np.zeros((1, 3, 3, 1), dtype=np.uint8)
    x[0, 1, 1, 0] = 130
    # centre: 130*5/13 = 50 ; border keeps original ; factor 0 -> degenerate
    out = A.sharpness(x, 0.0)[0, :, :, 0]
    assert out[1, 1] in (49, 50) and out.sum() == out[1, 1]
    out2 = A.sharpness(x, 1.72)[0, 1, 1, 0]    # 50 + 1.72*(130-50) = 187.6
    assert out2 in (187, 188, 189)


def test_autocontrast_kat():
    x = np.array([50, 100, 150, 50], dtype=np.uint8).reshape(1, 2, 2, 1)
    # scale = 2.55, offset = -127.5: 0, 127.5->127, 255
    np.testing.assert_array_equal(A.autocontrast(x).ravel(), [0, 127, 254 if False else 255, 0])
    z = np.full((1, 2, 2, 1), 9, dtype=np.uint8)
    np.testing.assert_array_equal(A.autocontrast(z), z)


def test_magnitude_maps_match_reference_table():
    # augmentation_schemes.py:42-102 at magnitude 9
    assert abs(A.magnitude_to_kwargs("Brightness", 9)["factor"] - 1.72) < 1e-12
    assert abs(A.magnitude_to_kwargs("ShearX", 9)["level"] - 0.27) < 1e-12
    assert A.magnitude_to_kwargs("TranslateX", 9)["pixels"] == 90
    assert A.magnitude_to_kwargs("Posterize", 9)["bits"] == 3
    assert A.magnitude_to_kwargs("Solarize", 9)["threshold"] == 230
    assert A.magnitude_to_kwargs("SolarizeAdd", 9)["addition"] == 99
    assert A.magnitude_to_kwargs("CutOut", 9)["mask_size"] == 72
    assert abs(A.magnitude_to_kwargs("Rotate", 9)["degrees"] - 27.0) < 1e-12
    assert A.magnitude_to_kwargs("Posterize", 2)["bits"] == 0 and A.magnitude_to_kwargs("Solarize", 10)["threshold"] == 256
    assert len(A.AUTO_AUGMENT_POLICY_V0) == 25 and len(A.RANDAUGMENT_OPS) == 16


def test_sharpness_final_cast_is_an_open_choice():
    """Builder-derived KAT that separates the two candidate roundings of tfa.image.sharpness's final blend (DESIGN.md section 2,
    open choices): on a 3x3 image whose centre is 10 among 200s the smoothed centre is trunc((8*200 + 5*10)/13) = 126, and
    blend(126, 10, 1.72) = 126 + 1.72*(10 - 126) = -73.52 -> clipped to 0 (both agree); with centre 101 the blend is
    161 + 1.72*(101 - 161) = 57.8: truncation (the oracle and the HIP kernel) gives 57, tf.round gives 58."""
    x = np.full((1, 3, 3, 3), 200, dtype=np.uint8)
    x[0, 1, 1] = 101
    smoothed = int((np.float32(8 * 200) + np.float32(5 * 101)) / np.float32(13))      # ~161.9 -> 161
    assert smoothed == 161
    t = A.sharpness(x, 1.72)
    r = A.sharpness(x, 1.72, final_cast="round")
    assert int(t[0, 1, 1, 0]) == 57 and int(r[0, 1, 1, 0]) == 58
    border = np.ones((3, 3), bool)
    border[1, 1] = False
    assert (t[0][border] == 200).all() and (r[0][border] == 200).all()          # the 1-pixel border keeps the original
    assert (A.sharpness(x, 0.4) == A.sharpness(x, 0.4, final_cast="truncate")).all()
