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
from .image_augmentations import (AutoContrast, Brightness, Color, Contrast, CutOut, Equalize, ImageNetNormalization, Invert,  # noqa: F401
                                  Posterize, RandomChance, RandomChoice, Rotate, Sharpness, ShearX, ShearY, Solarize, SolarizeAdd,
                                  TranslateX, TranslateY)
from .augmentation_schemes import AutoAugment, RandAugment  # noqa: F401
