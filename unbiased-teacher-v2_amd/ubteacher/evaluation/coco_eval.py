"""COCO bounding-box average precision (the metric behind the reference's accuracy tables, produced there by Detectron2's
COCOEvaluator / pycocotools - neither is in the reference tree nor installed, so this restates the published COCO protocol
[pycocotools-recall] and is pinned by known-answer tests, tests/test_evaluation.py):

  * per image and category, detections are taken in descending score order (at most 100 per image);
  * a detection is matched greedily, per IoU threshold t in {0.50, 0.55, ..., 0.95}, to the not-yet-matched ground truth of
    highest IoU >= t; crowd boxes may absorb any number of detections and use IoU = inter / area(det); matches to ignored
    ground truth (crowd, or outside the area range) make the detection ignored; unmatched detections outside the area
    range are ignored as well;
  * per category / area range, detections of all images are merged by score, precision is made monotonically
    non-increasing from the right and sampled at the 101 recall points 0, 0.01, ..., 1;
  * AP = mean over (IoU thresholds, recall points, categories with ground truth); AP50 / AP75 fix the threshold;
    APs / APm / APl restrict the area to (0, 32^2), [32^2, 96^2), [96^2, inf).
"""
import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
AREA_RNG = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}
MAX_DETS = 100


def _iou_matrix(dets, gts, crowd):
    """dets [D,4], gts [G,4] xyxy -> [D,G]; for crowd ground truth the union is the detection's area"""
    if len(dets) == 0 or len(gts) == 0:
        return np.zeros((len(dets), len(gts)))
    ad = (dets[:, 2] - dets[:, 0]) * (dets[:, 3] - dets[:, 1])
    ag = (gts[:, 2] - gts[:, 0]) * (gts[:, 3] - gts[:, 1])
    iw = np.clip(np.minimum(dets[:, None, 2], gts[None, :, 2]) - np.maximum(dets[:, None, 0], gts[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(dets[:, None, 3], gts[None, :, 3]) - np.maximum(dets[:, None, 1], gts[None, :, 1]), 0, None)
    inter = iw * ih
    union = np.where(crowd[None, :], ad[:, None], ad[:, None] + ag[None, :] - inter)
    return inter / np.maximum(union, 1e-12)


def _evaluate_image(dets, scores, gts, crowd, rng, garea=None):
    """one (image, category, area range): returns (scores, matched[T,D], ignored[T,D], number of non-ignored gt).
    garea: the annotations' `area` field when the ground truth comes from a COCO json (pycocotools ranks ground truth by that field -
    the segment area -, not by the box area); None = box areas"""
    order = np.argsort(-scores, kind="mergesort")[:MAX_DETS]
    dets, scores = dets[order], scores[order]
    if garea is None:
        garea = (gts[:, 2] - gts[:, 0]) * (gts[:, 3] - gts[:, 1]) if len(gts) else np.zeros(0)
    gignore = crowd | (garea < rng[0]) | (garea > rng[1])
    gorder = np.argsort(gignore, kind="mergesort")  # non-ignored first
    gts, crowd, gignore = gts[gorder], crowd[gorder], gignore[gorder]
    ious = _iou_matrix(dets, gts, crowd)
    T, D, G = len(IOU_THRS), len(dets), len(gts)
    dmatch = np.zeros((T, D), bool)
    dignore = np.zeros((T, D), bool)
    for ti, t in enumerate(IOU_THRS):
        gtaken = np.zeros(G, bool)
        for d in range(D):
            best, m = min(t, 1 - 1e-10), -1
            for g in range(G):
                if gtaken[g] and not crowd[g]:
                    continue
                if m > -1 and not gignore[m] and gignore[g]:
                    break  # a regular ground truth is already matched: do not trade it for an ignored one
                if ious[d, g] < best:
                    continue
                best, m = ious[d, g], g
            if m > -1:
                dmatch[ti, d] = True
                dignore[ti, d] = gignore[m]
                gtaken[m] = True
    darea = (dets[:, 2] - dets[:, 0]) * (dets[:, 3] - dets[:, 1]) if D else np.zeros(0)
    dout = (darea < rng[0]) | (darea > rng[1])
    dignore |= (~dmatch) & dout[None, :]
    return scores, dmatch, dignore, int((~gignore).sum())


def coco_box_ap(predictions, ground_truth, num_classes=None):
    """predictions: {image_id: dict(boxes [D,4] xyxy, scores [D], classes [D])};
    ground_truth: {image_id: dict(boxes [G,4], classes [G], iscrowd [G] optional)}.  Returns the six COCO numbers in percent
    (-1 where undefined, e.g. no ground truth of that size)."""
    cats = set()
    for g in ground_truth.values():
        cats.update(int(c) for c in np.asarray(g["classes"]).reshape(-1))
    if num_classes is not None:
        cats = {c for c in cats if 0 <= c < num_classes}
    cats = sorted(cats)
    precision = {k: -np.ones((len(IOU_THRS), len(REC_THRS), len(cats))) for k in AREA_RNG}
    for ci, c in enumerate(cats):
        for aname, rng in AREA_RNG.items():
            sc_all, dm_all, di_all, npig = [], [], [], 0
            for img, g in ground_truth.items():
                gb = np.asarray(g["boxes"], float).reshape(-1, 4)
                gc = np.asarray(g["classes"]).reshape(-1)
                gcrowd = np.asarray(g.get("iscrowd", np.zeros(len(gc))), bool).reshape(-1)
                gar = np.asarray(g["area"], float).reshape(-1) if g.get("area") is not None else None
                sel = gc == c
                p = predictions.get(img)
                if p is not None and len(np.asarray(p["classes"]).reshape(-1)):
                    pc = np.asarray(p["classes"]).reshape(-1)
                    psel = pc == c
                    pb = np.asarray(p["boxes"], float).reshape(-1, 4)[psel]
                    ps = np.asarray(p["scores"], float).reshape(-1)[psel]
                else:
                    pb, ps = np.zeros((0, 4)), np.zeros(0)
                if not sel.any() and len(pb) == 0:
                    continue
                s, dm, di, n = _evaluate_image(pb, ps, gb[sel], gcrowd[sel], rng, None if gar is None else gar[sel])
                sc_all.append(s); dm_all.append(dm); di_all.append(di); npig += n
            if npig == 0:
                continue
            scores = np.concatenate(sc_all) if sc_all else np.zeros(0)
            order = np.argsort(-scores, kind="mergesort")
            dm = np.concatenate(dm_all, axis=1)[:, order] if dm_all else np.zeros((len(IOU_THRS), 0), bool)
            di = np.concatenate(di_all, axis=1)[:, order] if di_all else np.zeros((len(IOU_THRS), 0), bool)
            tps = np.cumsum(dm & ~di, axis=1).astype(float)
            fps = np.cumsum(~dm & ~di, axis=1).astype(float)
            for ti in range(len(IOU_THRS)):
                tp, fp = tps[ti], fps[ti]
                rc = tp / npig
                pr = tp / np.maximum(tp + fp, np.spacing(1))
                for i in range(len(pr) - 1, 0, -1):  # monotone envelope
                    if pr[i] > pr[i - 1]:
                        pr[i - 1] = pr[i]
                inds = np.searchsorted(rc, REC_THRS, side="left")
                q = np.zeros(len(REC_THRS))
                ok = inds < len(pr)
                q[ok] = pr[inds[ok]]
                precision[aname][ti, :, ci] = q

    def mean_ap(aname, ti=None):
        p = precision[aname] if ti is None else precision[aname][ti:ti + 1]
        p = p[p > -1]
        return float(p.mean() * 100.0) if p.size else -1.0

    return {"AP": mean_ap("all"), "AP50": mean_ap("all", 0), "AP75": mean_ap("all", 5),
            "APs": mean_ap("small"), "APm": mean_ap("medium"), "APl": mean_ap("large")}


class COCOBoxEvaluator:
    """DatasetEvaluator surface (reset / process / evaluate) on in-memory ground truth: every input dict carries `image_id`
    and `instances` (gt_boxes, gt_classes) at the ORIGINAL image size (`height`, `width`); outputs are the model's eval-mode
    results `{"instances": Instances(pred_boxes, scores, pred_classes)}` already rescaled by detector_postprocess."""

    def __init__(self, num_classes=None, dataset_name=None):
        """dataset_name: a registered dataset (DatasetCatalog) whose dicts carry the ground truth - Detectron2's COCOEvaluator(dataset_name)
        reads it from the set's json [D2-recall]; the test mapper drops `annotations` from its inputs.  None: the ground truth rides on
        every input as `instances` (the synthetic test loader)."""
        self.num_classes = num_classes
        self._dataset_gt = None
        if dataset_name is not None:
            from ..data import DatasetCatalog
            from ..data.dataset_mapper import to_xyxy_abs
            self._dataset_gt = {}
            for d in DatasetCatalog.get(dataset_name):
                annos = d.get("annotations", [])
                self._dataset_gt[d["image_id"]] = dict(
                    boxes=np.asarray([to_xyxy_abs(a) for a in annos], float).reshape(-1, 4),
                    classes=np.asarray([a["category_id"] for a in annos], np.int64),
                    iscrowd=np.asarray([a.get("iscrowd", 0) for a in annos], bool),
                    area=np.asarray([a["area"] for a in annos], float) if annos and all("area" in a for a in annos) else None)
        self.reset()

    def reset(self):
        self._pred, self._gt = {}, {}

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            iid = inp["image_id"]
            if self._dataset_gt is not None:
                self._gt[iid] = self._dataset_gt[iid]
                inst = out["instances"] if "instances" in out else out["proposals"]
                self._pred[iid] = dict(boxes=inst.pred_boxes.tensor.detach().cpu().numpy(), scores=inst.scores.detach().cpu().numpy(),
                                       classes=inst.pred_classes.detach().cpu().numpy())
                continue
            gt = inp["instances"]
            self._gt[iid] = dict(boxes=gt.gt_boxes.tensor.detach().cpu().numpy(), classes=gt.gt_classes.detach().cpu().numpy())
            inst = out["instances"] if "instances" in out else out["proposals"]
            self._pred[iid] = dict(boxes=inst.pred_boxes.tensor.detach().cpu().numpy(), scores=inst.scores.detach().cpu().numpy(),
                                   classes=inst.pred_classes.detach().cpu().numpy())

    def evaluate(self):
        """Detectron2 COCOEvaluator(distributed=True).evaluate [D2-recall]: the test loader shards the set over the ranks
        (build_detection_test_loader + InferenceSampler), so every rank's predictions (and the ground truth that rode on its inputs) are
        gathered on the main rank, merged by image id, and scored there; the other ranks return {}."""
        from ..utils import comm
        pred, gt = self._pred, self._gt
        if comm.get_world_size() > 1:
            comm.synchronize()
            parts = comm.gather((pred, gt), dst=0)
            if not comm.is_main_process():
                return {}
            pred, gt = {}, {}
            for p, g in parts:
                pred.update(p)
                gt.update(g)
        return {"bbox": coco_box_ap(pred, gt, self.num_classes)}
