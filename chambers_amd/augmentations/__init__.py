"""chambers.augmentations namespace (reference: chambers/augmentations/__init__.py:14-39).
The Keras preprocessing layers the reference re-exports (:1-13) are not chambers code and are
out of scope of this build."""
from .image_augmentations import (  # noqa: F401
    ImageNetNormalization,
    RandomChoice,
    RandomChance,
    AutoContrast,
    Equalize,
    Invert,
    Rotate,
    Posterize,
    Solarize,
    SolarizeAdd,
    Color,
    Contrast,
    Brightness,
    Sharpness,
    ShearX,
    ShearY,
    TranslateX,
    TranslateY,
    CutOut,
)
from .augmentation_schemes import (  # noqa: F401
    AutoAugment,
    RandAugment,
)
