"""FCOS on the texture toy set: is the teacher, right after the burn-in copy, the student?  (the evaluation hooks showed teacher AP 0.0 while the
student had 17-59)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, numpy as np, random
import make_tiny_coco, make_synthetic_backbone
from ubteacher.data import register_coco_instances
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config
kind = sys.argv[1] if len(sys.argv) > 1 else "fcos"
root = "/tmp/tcd_ds"
sys.argv = ["x", root, "64", "16"]; make_tiny_coco.main()
w = "/tmp/tcd_bb.pth"; sys.argv = ["x", kind, w, "0"]; make_synthetic_backbone.main()
for split in ("train", "val"):
    register_coco_instances("tcd_" + split, {}, os.path.join(root, "coco/annotations/instances_%s2017.json" % split), os.path.join(root, "coco/%s2017" % split))
cfg = get_config(kind, 1, ["MODEL.DEVICE", "cuda", "MODEL.WEIGHTS", w, "SOLVER.MAX_ITER", 2000, "SEMISUPNET.BURN_UP_STEP", 300,
                           "SOLVER.IMG_PER_BATCH_LABEL", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 4, "SOLVER.CHECKPOINT_PERIOD", 0, "TEST.EVAL_PERIOD", 0,
                           "DATALOADER.SUP_PERCENT", 50.0, "DATALOADER.RANDOM_DATA_SEED_PATH", os.path.join(root, "seed.json"),
                           "INPUT.MIN_SIZE_TRAIN", (160, 224), "INPUT.MAX_SIZE_TRAIN", 320, "INPUT.MIN_SIZE_TEST", 192, "INPUT.MAX_SIZE_TEST", 320, "OUTPUT_DIR", ""])
cfg.DATASETS.TRAIN = ("tcd_train",); cfg.DATASETS.TEST = ("tcd_val",)
torch.manual_seed(1); np.random.seed(1); random.seed(1)
T = UBTeacherTrainer if kind == "fcos" else UBRCNNTeacherTrainer
tr = T(cfg); tr.resume_or_load(resume=False); tr.checkpointer.save = lambda *a, **k: None
def diff(tag):
    s, t = tr.model.state_dict(), tr.model_teacher.state_dict()
    worst = sorted(((float((s[k].float() - t[k].float()).abs().max()), k) for k in s), reverse=True)[:4]
    print(tag, "largest student-teacher differences:", [(round(a, 5), k) for a, k in worst])
def ap(tag):
    rs = T.test(cfg, tr.model)["bbox"]; rt = T.test(cfg, tr.model_teacher)["bbox"]
    print(tag, "student AP %.1f AP50 %.1f | teacher AP %.1f AP50 %.1f" % (rs["AP"], rs["AP50"], rt["AP"], rt["AP50"]))
tr.train_loop(0, 300); diff("after 300 burn-in iterations:"); ap("after 300:")
amp = lambda: [round(v, 1) for v in tr._amp_state.cpu().tolist()] if getattr(tr, "_amp_state", None) is not None else None
print("loss-scale state {scale, found_inf, clean steps} before the boundary:", amp())
tr.train_loop(300, 301); diff("after the boundary step:"); ap("after 301:"); print("loss-scale state after the boundary step:", amp())
for a, b in ((301, 302), (302, 303), (303, 310), (310, 340)):
    tr.train_loop(a, b); print("loss-scale state after iteration %d:" % (b - 1), amp())
diff("after 340:"); ap("after 340:")
