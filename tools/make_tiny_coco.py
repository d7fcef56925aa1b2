"""A COCO-layout dataset tree small enough to write in a second (textured patches on noise: 4 categories by TEXTURE), for driving the launcher's REAL data path end to end (registration by
$DETECTRON2_DATASETS, label / unlabel split by a seed table, two-crop mapper, evaluation on files):
  ROOT/coco/train2017/*.png, ROOT/coco/val2017/*.png, ROOT/coco/annotations/instances_{train,val}2017.json, ROOT/seed.json
(seed.json: the reference's dataseed/COCO_supervision.txt shape {"<percent>": {"<seed>": [labeled indices]}} for 12.5 / 25 / 50 % of the train images)
usage: python tools/make_tiny_coco.py ROOT [n_train] [n_val] [texture|colour]"""
import json
import os
import sys

import numpy as np
from PIL import Image

CODE = "texture"      # how a category looks: "texture" (default) or "colour" (4th argument)
CATS = [{"id": 1, "name": "person"}, {"id": 18, "name": "dog"}, {"id": 44, "name": "bottle"}, {"id": 90, "name": "toothbrush"}]


def write_split(root, split, n, rng, first_id):
    img_dir = os.path.join(root, "coco", split)
    os.makedirs(img_dir, exist_ok=True)
    images, annos = [], []
    for i in range(n):
        h, w = [(120, 160), (144, 128), (96, 192), (128, 128)][i % 4]
        fn = "%012d.png" % (first_id + i)
        px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        images.append({"id": first_id + i, "file_name": fn, "height": h, "width": w})
        for _ in range(int(rng.integers(1, 4))):
            bw, bh = float(rng.uniform(16, w / 2)), float(rng.uniform(16, h / 2))
            x, y = float(rng.uniform(0, w - bw)), float(rng.uniform(0, h - bh))
            c = int(rng.integers(0, 4))
            if CODE == "colour":   # a flat patch per category: learned in a few hundred supervised iterations (tests/test_learning_gpu.py)
                px[int(y):int(y + bh), int(x):int(x + bw)] = (60 * c + 30, 255 - 60 * c, 40 * c)
                annos.append({"id": len(annos) + 1 + 100000 * (split == "val2017"), "image_id": first_id + i, "category_id": CATS[c]["id"],
                              "bbox": [x, y, bw, bh], "area": bw * bh, "iscrowd": 0})
                continue
            # a TEXTURE per category in two random colours (horizontal stripes / vertical stripes / checkerboard / flat): learnable, and
            # - unlike a colour code - still the same category under the strong view's colour jitter and grayscale (the teacher / student
            # phase needs that: a pseudo label from the weak view must still be true of the strong view)
            yy, xx = np.mgrid[int(y):int(y + bh), int(x):int(x + bw)]
            m = [(yy // 6) % 2, (xx // 6) % 2, ((yy // 6) + (xx // 6)) % 2, np.zeros_like(yy)][c].astype(bool)
            a, b = rng.integers(0, 256, 3), rng.integers(0, 256, 3)
            while np.abs(a.astype(int) - b.astype(int)).sum() < 200:
                b = rng.integers(0, 256, 3)
            px[int(y):int(y + bh), int(x):int(x + bw)] = np.where(m[..., None], a, b).astype(np.uint8)
            annos.append({"id": len(annos) + 1 + 100000 * (split == "val2017"), "image_id": first_id + i, "category_id": CATS[c]["id"],
                          "bbox": [x, y, bw, bh], "area": bw * bh, "iscrowd": 0})
        Image.fromarray(px, "RGB").save(os.path.join(img_dir, fn))
    os.makedirs(os.path.join(root, "coco", "annotations"), exist_ok=True)
    with open(os.path.join(root, "coco", "annotations", "instances_%s.json" % split), "w") as f:
        json.dump({"images": images, "annotations": annos, "categories": CATS}, f)


def main():
    global CODE
    CODE = sys.argv[4] if len(sys.argv) > 4 else "texture"
    assert CODE in ("texture", "colour")
    root = sys.argv[1]
    n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    n_val = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    rng = np.random.default_rng(0)
    write_split(root, "train2017", n_train, rng, 1000)
    write_split(root, "val2017", n_val, rng, 5000)
    perm = [int(i) for i in rng.permutation(n_train)]
    idx = sorted(perm[: n_train // 2])
    table = {"50.0": {"0": idx, "1": idx[::-1]}}
    for pct in (12.5, 25.0):       # nested subsets of the 50 % split
        sub = sorted(perm[: int(pct / 100.0 * n_train)])
        table[str(pct)] = {"0": sub, "1": sub[::-1]}
    with open(os.path.join(root, "seed.json"), "w") as f:
        json.dump(table, f)
    print("wrote", root, n_train, "train /", n_val, "val images; labeled indices", idx)


if __name__ == "__main__":
    main()
