"""Launcher with the reference's CLI shape (train_net.py:15-73): --config-file, --num-gpus, --resume, --eval-only,
KEY VALUE overrides.  `--num-gpus N` starts N ranks, one per GPU (ubteacher.engine.launch, the counterpart of the
`launch(main, args.num_gpus, ...)` call at train_net.py:62-73); under `python -m torch.distributed.run --nproc-per-node N`
the process joins that world instead of spawning."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))

from ubteacher import add_ubteacher_config  # noqa: E402
from ubteacher.d2 import get_cfg  # noqa: E402
from ubteacher.engine import UBRCNNTeacherTrainer, UBTeacherTrainer  # noqa: E402
from ubteacher.engine.launch import dist_info, launch  # noqa: E402


def setup(args):
    cfg = get_cfg()
    add_ubteacher_config(cfg)
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    if cfg.MODEL.DEVICE == "cuda":  # one process per GPU: this rank's device
        cfg.MODEL.DEVICE = "cuda:%d" % dist_info().get("device", 0)
    cfg.freeze()
    setup_logger(cfg.OUTPUT_DIR, dist_info().get("rank", 0))
    seed_all_rng(None if cfg.SEED < 0 else cfg.SEED + dist_info().get("rank", 0))
    return cfg


def seed_all_rng(seed=None):
    """Detectron2's default_setup seeds Python / NumPy / torch with cfg.SEED + rank (a fresh seed per process, logged, when SEED < 0) [D2-recall]"""
    import logging
    import random

    import numpy as np
    import torch
    if seed is None:
        seed = (os.getpid() + int.from_bytes(os.urandom(3), "big")) % (2 ** 31)
        logging.getLogger(__name__).info("Using a generated random seed {}".format(seed))
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)


def setup_logger(output_dir, rank):
    """what the reference gets from Detectron2's default_setup (train_net.py:31): INFO lines of the main process on stdout and in
    OUTPUT_DIR/log.txt (other ranks: OUTPUT_DIR/log.txt.rank<N>, warnings only on their consoles)"""
    import logging
    root = logging.getLogger()
    if getattr(root, "_utv2_configured", False):
        return
    root._utv2_configured = True
    root.setLevel(logging.INFO)
    fmt = logging.Formatter("[%(asctime)s %(name)s]: %(message)s", datefmt="%m/%d %H:%M:%S")
    ch = logging.StreamHandler(sys.stdout)
    ch.setLevel(logging.INFO if rank == 0 else logging.WARNING)
    ch.setFormatter(fmt)
    root.addHandler(ch)
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
        fh = logging.FileHandler(os.path.join(output_dir, "log.txt" if rank == 0 else "log.txt.rank%d" % rank))
        fh.setLevel(logging.INFO)
        fh.setFormatter(fmt)
        root.addHandler(fh)


def main(args):
    cfg = setup(args)
    if cfg.SEMISUPNET.Trainer == "ubteacher":
        Trainer = UBTeacherTrainer
    elif cfg.SEMISUPNET.Trainer == "ubteacher_rcnn":
        Trainer = UBRCNNTeacherTrainer
    else:
        raise ValueError("Trainer Name is not found.")

    if args.eval_only:
        # reference train_net.py:37-54: FCOS evaluates the TEACHER of the teacher/student checkpoint, Faster-RCNN a plain model
        from ubteacher.checkpoint import DetectionTSCheckpointer
        from ubteacher.modeling.ts_ensemble import EnsembleTSModel
        model = Trainer.build_model(cfg)
        if cfg.SEMISUPNET.Trainer == "ubteacher":
            model_teacher = Trainer.build_model(cfg)
            ensem_ts_model = EnsembleTSModel(model_teacher, model)
            DetectionTSCheckpointer(ensem_ts_model, save_dir=cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
            return Trainer.test(cfg, ensem_ts_model.modelTeacher)
        from ubteacher.checkpoint import DetectionCheckpointer
        DetectionCheckpointer(model, save_dir=cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
        return Trainer.test(cfg, model)

    trainer = Trainer(cfg)
    trainer.resume_or_load(resume=args.resume)
    return trainer.train()


def default_argument_parser():
    """the arguments of Detectron2's default_argument_parser that the reference launcher uses [D2-recall]"""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default=os.path.join(ROOT, "configs", "utv2_fcos_r50.yaml"), metavar="FILE")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--num-gpus", type=int, default=1, help="number of gpus *per machine*")
    ap.add_argument("--num-machines", type=int, default=1)
    ap.add_argument("--machine-rank", type=int, default=0)
    ap.add_argument("--dist-url", default="auto")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    return ap


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    print("Command Line Args:", args)
    launch(main, args.num_gpus, num_machines=args.num_machines, machine_rank=args.machine_rank, dist_url=args.dist_url, args=(args,))
