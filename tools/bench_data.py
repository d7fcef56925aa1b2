"""Throughput of the two-crop mapper: GPU mapper (csrc/augment.hip) vs the same decisions executed by Pillow on one
host core (the code the reference pipeline runs per dataloader worker).  COCO-shaped 480x640-class images, INPUT.MIN_SIZE_TRAIN
(400, 1200) range / max 1333 as in the shipped FCOS configs.  Prints images/s (one image = both views) and per-kernel algorithmic bytes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import numpy as np
import torch
from ubteacher.presets import get_config
from ubteacher.data import DatasetMapperTwoCropSeparate, synthetic_coco_dicts

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = get_config("fcos", 1, ["MODEL.DEVICE", "cuda"])
cfg.INPUT.MIN_SIZE_TRAIN = (400, 1200)
cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING = "range"
cfg.SEED = 5
dicts = synthetic_coco_dicts(n, seed=9)
mapper = DatasetMapperTwoCropSeparate(cfg, True)
for d in dicts[:8]:
    mapper(d)
torch.cuda.synchronize()
params, outpix = [], 0
t0 = time.perf_counter()
for d in dicts:
    s, w = mapper(d)
    params.append(mapper.last_params)
    outpix += s["image"].shape[1] * s["image"].shape[2]
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
print("gpu mapper: %.1f images/s (%.2f ms/image, both views; %d images, mean output %.2f Mpx)" % (n / t_gpu, 1e3 * t_gpu / n, n, outpix / n / 1e6))

# the same decisions through Pillow on one core
from PIL import Image, ImageEnhance, ImageFilter
m = min(n, 24)
rng = np.random.default_rng(0)
t0 = time.perf_counter()
for d, p in zip(dicts[:m], params[:m]):
    im = Image.fromarray(np.ascontiguousarray(d["image"][:, :, ::-1])).resize((p["neww"], p["newh"]), Image.BILINEAR)
    if p["flip"]:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    weak = np.asarray(im)
    x = im
    if p["jitter"]:
        for fn in p["order"]:
            if fn == 0: x = ImageEnhance.Brightness(x).enhance(p["brightness"])
            elif fn == 1: x = ImageEnhance.Contrast(x).enhance(p["contrast"])
            elif fn == 2: x = ImageEnhance.Color(x).enhance(p["saturation"])
            else:
                h, s_, v = x.convert("HSV").split()
                nh = (np.array(h, dtype=np.uint8).astype(np.int32) + (int(np.trunc(p["hue"] * 255)) & 255)).astype(np.uint8)
                x = Image.merge("HSV", (Image.fromarray(nh, "L"), s_, v)).convert("RGB")
    if p["gray"]:
        x = Image.fromarray(np.repeat(np.asarray(x.convert("L"))[..., None], 3, axis=2))
    if p["blur"]:
        x = x.filter(ImageFilter.GaussianBlur(radius=p["sigma"]))
    t = torch.from_numpy(np.asarray(x).copy()).permute(2, 0, 1).float().div(255)
    for r in p["erase"]:
        if r is not None:
            t[:, r[0]:r[0] + r[2], r[1]:r[1] + r[3]] = torch.empty((3, r[2], r[3])).normal_()
    strong = t.mul(255).byte()
t_cpu = time.perf_counter() - t0
print("pillow, 1 core: %.1f images/s (%.1f ms/image; %d images)" % (m / t_cpu, 1e3 * t_cpu / m, m))
print("ratio: %.1fx one core" % ((n / t_gpu) / (m / t_cpu)))
