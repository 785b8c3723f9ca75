"""chambers.augmentations namespace (reference: chambers/augmentations/__init__.py:14-39).
Of the Keras preprocessing layers the reference re-exports (:1-13), the input-side ones of SURVEY §8f rank 3 are built
(Resizing, CenterCrop, RandomCrop, RandomFlip, Rescaling — .preprocessing); RandomRotation / RandomZoom / RandomTranslation /
RandomContrast / RandomHeight / RandomWidth are not."""
from .preprocessing import (  # noqa: F401
    CenterCrop,
    RandomCrop,
    RandomFlip,
    Rescaling,
    Resizing,
    ResizingMinMax,
)
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
