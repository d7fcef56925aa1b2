from .build import (InferenceSampler, TrainingSampler, build_detection_semisup_train_loader_two_crops,  # noqa: F401
                    build_detection_test_loader, build_semisup_batch_data_loader_two_crop, divide_label_unlabel, get_detection_dataset_dicts)
from .common import AspectRatioGroupedSemiSupDatasetTwoCrop, MapDataset  # noqa: F401
from .dataset_mapper import DatasetMapperTwoCropSeparate  # noqa: F401
from .datasets import (DatasetCatalog, MetadataCatalog, load_coco_json, register_coco_instances,  # noqa: F401
                       register_coco_unlabel_instances, register_synthetic, synthetic_coco_dicts)
