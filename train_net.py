"""Launcher with the reference's CLI shape (train_net.py:15-73): --config-file, --num-gpus, --resume,
KEY VALUE overrides.  One process per GPU; for --num-gpus > 1 start it under
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train_net.py ...`."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ubteacher import add_ubteacher_config  # noqa: E402
from ubteacher.d2 import get_cfg  # noqa: E402
from ubteacher.engine import UBRCNNTeacherTrainer, UBTeacherTrainer  # noqa: E402


def setup(args):
    cfg = get_cfg()
    add_ubteacher_config(cfg)
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if cfg.MODEL.DEVICE == "cuda":
        cfg.MODEL.DEVICE = "cuda:%d" % local_rank
    cfg.freeze()
    return cfg


def main(args):
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    cfg = setup(args)
    if cfg.SEMISUPNET.Trainer == "ubteacher":
        Trainer = UBTeacherTrainer
    elif cfg.SEMISUPNET.Trainer == "ubteacher_rcnn":
        Trainer = UBRCNNTeacherTrainer
    else:
        raise ValueError("Trainer Name is not found.")
    if args.eval_only:
        raise NotImplementedError("--eval-only: COCO evaluation is outside the training-step scope (SURVEY 8f)")
    trainer = Trainer(cfg)
    trainer.resume_or_load(resume=args.resume)
    return trainer.train()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default=os.path.join(ROOT, "configs", "utv2_fcos_r50.yaml"))
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    main(ap.parse_args())
