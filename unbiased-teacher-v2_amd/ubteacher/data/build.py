"""Loader construction of the semi-supervised trainers (reference ubteacher/data/build.py:30-53,144-272)."""
import itertools
import json

import numpy as np
import torch

from ..utils import comm
from .common import AspectRatioGroupedSemiSupDatasetTwoCrop, MapDataset
from .datasets import DatasetCatalog


def divide_label_unlabel(dataset_dicts, SupPercent, random_data_seed, random_data_seed_path):
    """Split by the pre-generated index lists of dataseed/COCO_supervision.txt: {str(percent): {str(seed): [indices]}} (build.py:30-53).
    Order inside each half follows the dataset order; the listed count must equal int(percent / 100 * len)."""
    num_all = len(dataset_dicts)
    num_label = int(SupPercent / 100.0 * num_all)
    with open(random_data_seed_path, "r") as f:
        table = json.load(f)
    labeled_idx = np.array(table[str(SupPercent)][str(random_data_seed)])
    assert labeled_idx.shape[0] == num_label, "Number of READ_DATA is mismatched."
    chosen = set(int(i) for i in labeled_idx)
    label_dicts = [d for i, d in enumerate(dataset_dicts) if i in chosen]
    unlabel_dicts = [d for i, d in enumerate(dataset_dicts) if i not in chosen]
    return label_dicts, unlabel_dicts


class TrainingSampler:
    """Detectron2 TrainingSampler: an infinite stream of (shuffled) permutations of range(size) from a seed shared by all ranks; rank r
    takes elements r, r + world, r + 2 world, ... of that stream."""

    def __init__(self, size, shuffle=True, seed=None, rank=None, world_size=None):
        assert size > 0
        self._size, self._shuffle = size, shuffle
        self._seed = int(comm.shared_random_seed() if seed is None else seed)
        self._rank = comm.get_rank() if rank is None else rank
        self._world_size = comm.get_world_size() if world_size is None else world_size

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from range(self._size)

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)


def build_semisup_batch_data_loader_two_crop(dataset, sampler, total_batch_size_label, total_batch_size_unlabel, *,
                                             aspect_ratio_grouping=False, num_workers=0, mapper=None):
    """dataset / sampler: (labeled, unlabeled) pairs of dict lists and index samplers (build.py:216-272).  num_workers is accepted for
    config compatibility; the mapper runs on the GPU in this process (decode is the only host work)."""
    world_size = comm.get_world_size()
    assert total_batch_size_label > 0 and total_batch_size_label % world_size == 0, \
        "Total label batch size ({}) must be divisible by the number of gpus ({}).".format(total_batch_size_label, world_size)
    assert total_batch_size_unlabel > 0 and total_batch_size_unlabel % world_size == 0, \
        "Total unlabel batch size ({}) must be divisible by the number of gpus ({}).".format(total_batch_size_label, world_size)
    if not aspect_ratio_grouping:
        raise NotImplementedError("ASPECT_RATIO_GROUPING = False is not supported yet")
    label_dicts, unlabel_dicts = dataset
    label_sampler, unlabel_sampler = sampler
    return AspectRatioGroupedSemiSupDatasetTwoCrop(
        (MapDataset(label_dicts, mapper, label_sampler), MapDataset(unlabel_dicts, mapper, unlabel_sampler)),
        (total_batch_size_label // world_size, total_batch_size_unlabel // world_size))


def get_detection_dataset_dicts(names, filter_empty=True):
    if isinstance(names, str):
        names = [names]
    assert len(names), names
    dicts = list(itertools.chain.from_iterable(DatasetCatalog.get(n) for n in names))
    if filter_empty and len(dicts) and "annotations" in dicts[0]:
        dicts = [d for d in dicts if any(a.get("iscrowd", 0) == 0 for a in d.get("annotations", []))]
    assert len(dicts), "No valid data found in {}.".format(",".join(names))
    return dicts


def build_detection_semisup_train_loader_two_crops(cfg, mapper=None):
    """build.py:144-213: label / unlabel split (cross-dataset, or one dataset divided by DATALOADER.SUP_PERCENT), two samplers, the
    two-crop mapper and the aspect-ratio grouped 4-tuple batcher."""
    if cfg.DATASETS.CROSS_DATASET:
        label_dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN_LABEL, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
        unlabel_dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN_UNLABEL, filter_empty=False)
    else:
        dataset_dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
        label_dicts, unlabel_dicts = divide_label_unlabel(dataset_dicts, cfg.DATALOADER.SUP_PERCENT, cfg.DATALOADER.RANDOM_DATA_SEED,
                                                          cfg.DATALOADER.RANDOM_DATA_SEED_PATH)
    if mapper is None:
        from .dataset_mapper import DatasetMapperTwoCropSeparate
        mapper = DatasetMapperTwoCropSeparate(cfg, True)
    sampler_name = cfg.DATALOADER.SAMPLER_TRAIN
    if sampler_name == "TrainingSampler":
        label_sampler = TrainingSampler(len(label_dicts))
        unlabel_sampler = TrainingSampler(len(unlabel_dicts))
    elif sampler_name == "RepeatFactorTrainingSampler":
        raise NotImplementedError("{} not yet supported.".format(sampler_name))
    else:
        raise ValueError("Unknown training sampler: {}".format(sampler_name))
    return build_semisup_batch_data_loader_two_crop(
        (label_dicts, unlabel_dicts), (label_sampler, unlabel_sampler), cfg.SOLVER.IMG_PER_BATCH_LABEL, cfg.SOLVER.IMG_PER_BATCH_UNLABEL,
        aspect_ratio_grouping=cfg.DATALOADER.ASPECT_RATIO_GROUPING, num_workers=cfg.DATALOADER.NUM_WORKERS, mapper=mapper)


class InferenceSampler:
    """Detectron2 InferenceSampler [D2-recall]: the indices 0..size-1 cut into world_size contiguous shards whose sizes differ by at
    most one (the first size % world shards take the extra element); rank r iterates its shard in order."""

    def __init__(self, size, rank=None, world_size=None):
        assert size > 0
        rank = comm.get_rank() if rank is None else rank
        world = comm.get_world_size() if world_size is None else world_size
        shard, left = size // world, size % world
        sizes = [shard + int(r < left) for r in range(world)]
        begin = sum(sizes[:rank])
        self._indices = range(begin, min(begin + sizes[rank], size))

    def __iter__(self):
        yield from self._indices

    def __len__(self):
        return len(self._indices)


class DetectionTestLoader:
    """fixed-length loader of the evaluation path: batches of ONE mapped image (Detectron2 build_detection_test_loader: batch size 1,
    InferenceSampler, trivial collate [D2-recall]).  Mapped lazily: the decode + resize of image i happens when batch i is asked for."""

    def __init__(self, dicts, mapper, sampler):
        self.dicts, self.mapper, self.sampler = dicts, mapper, sampler

    def __len__(self):
        return len(self.sampler)

    def __iter__(self):
        for idx in self.sampler:
            yield [self.mapper(self.dicts[idx])]


def build_detection_test_loader(cfg, dataset_name, mapper=None):
    """Detectron2 build_detection_test_loader(cfg, name) [D2-recall] as the reference's trainers call it (engine/trainer.py:554-608 via
    DefaultTrainer.build_test_loader): every image of the set (no filtering), the test-time mapper (ResizeShortestEdge(MIN_SIZE_TEST,
    MAX_SIZE_TEST), no flip; the dict keeps the ORIGINAL height / width and image_id, annotations are dropped), one image per batch,
    the set sharded over the ranks."""
    dicts = get_detection_dataset_dicts(dataset_name, filter_empty=False)
    if mapper is None:
        from .dataset_mapper import DatasetMapperTwoCropSeparate
        mapper = DatasetMapperTwoCropSeparate(cfg, is_train=False)
    return DetectionTestLoader(dicts, mapper, InferenceSampler(len(dicts)))
