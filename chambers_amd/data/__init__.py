"""Input pipeline in front of the augmentation path (reference: chambers/data/)."""
from .dataset import (Dataset, InterleaveImageClassDataset, InterleaveImageClassTripletDataset, InterleaveImageTripletDataset,  # noqa: F401
                      SequentialImageDataset, set_n_parallel)
from .io import match_img_files, match_img_files_triplet, match_nested_set, read_and_decode_image  # noqa: F401
