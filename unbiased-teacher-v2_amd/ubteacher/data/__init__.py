from .build import (TrainingSampler, build_detection_semisup_train_loader_two_crops, build_semisup_batch_data_loader_two_crop,  # noqa: F401
                    divide_label_unlabel, get_detection_dataset_dicts)
from .common import AspectRatioGroupedSemiSupDatasetTwoCrop, MapDataset  # noqa: F401
from .dataset_mapper import DatasetMapperTwoCropSeparate  # noqa: F401
from .datasets import DatasetCatalog, register_synthetic, synthetic_coco_dicts  # noqa: F401
