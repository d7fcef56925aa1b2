"""Where does the coupled Faster-RCNN step (tests/test_rcnn_step_gpu.py::test_rcnn_full_semisup_step_parity) deviate from the oracle?
Prints the product's teacher pseudo boxes against the oracle's (per coordinate), and the per-loss deviations with the oracle's own and
with the product's pseudo boxes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from oracle import utv2_oracle as O
from tests.utv2_testutil import FixedLoader, cpu_state, make_batch, rcnn_tune as tune
from tests.test_rcnn_step_gpu import rcnn_cfg, H, W
from ubteacher.engine import UBRCNNTeacherTrainer
torch.set_printoptions(precision=7, linewidth=200)
cfg = rcnn_cfg()
torch.manual_seed(0)
prod, orac = make_batch(31, 2, 2, H, W, "cuda")
tr = UBRCNNTeacherTrainer(cfg, data_loader=FixedLoader(prod))
mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1); pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
sd_s = tune(cpu_state(tr.model), [d["image"] for d in orac[3]], mean, pstd)
sd_t = dict(sd_s); sd_t["roi_heads.box_predictor.bbox_pred_std.bias"] = torch.full((4,), -3.0)
tr.model.load_state_dict(sd_s); tr.model_teacher.load_state_dict(sd_t)
tr.iter = 1; tr.optimizer.param_groups[0]["lr"] = 0.01
g = torch.Generator().manual_seed(99)
rpn_keys, roi_keys = [], []
def rpn_src(n, m, device):
    k = torch.rand(n, m, generator=g); rpn_keys.append(k); return k.to(device)
def roi_src(n, m, device):
    k = torch.rand(n, m, generator=g); roi_keys.append(k); return k.to(device)
tr.model.proposal_generator.sample_keys = rpn_src; tr.model.roi_heads.sample_keys = roi_src
tr.run_step_full_semisup(); rec = tr.flush_metrics(); torch.cuda.synchronize()
post = cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN
t_sd = O.ema_update(sd_s, sd_t, cfg.SEMISUPNET.EMA_KEEP_RATE)
with torch.no_grad():
    pseudo, tprops = O.rcnn_teacher(t_sd, [d["image"] for d in orac[3]], mean, pstd, thr=cfg.SEMISUPNET.BBOX_THRESHOLD)
gl = tr._last_pseudo
prod_pseudo = []
for i, p in enumerate(pseudo):
    m = gl["valid"][i].bool()
    pb = gl["boxes"][i][m].cpu(); ps = gl["scores"][i][m].cpu(); pc = gl["classes"][i][m].cpu().long(); pstd_ = gl["pred_boxes_std"][i][m].cpu()
    print("image", i, "oracle", len(p["boxes"]), "product", len(pb))
    if len(pb) == len(p["boxes"]):
        print(" box abs diff max %.3e   score diff max %.3e  class equal %s  std diff %.3e" % (float((pb - p["boxes"]).abs().max()), float((ps - p["scores"]).abs().max()),
              bool((pc == p["classes"]).all()), float((pstd_ - p["pred_boxes_std"]).abs().max())))
        print(" per-box max diff", (pb - p["boxes"]).abs().max(dim=1)[0])
        print(" oracle boxes", p["boxes"]); print(" product boxes", pb)
    prod_pseudo.append(dict(boxes=pb, classes=pc, scores=ps, pred_boxes_std=pstd_))
def compact_roi(keys, nprops, ngts):
    return [torch.cat((keys[i, :nprops[i]], keys[i, post:post + ngts[i]])) for i in range(keys.shape[0])]
for name, ps in (("oracle pseudo", pseudo), ("product pseudo", prod_pseudo)):
    with torch.no_grad():
        _, props_sup, _ = O.rcnn_student_losses(sd_s, [d["image"] for d in orac[0] + orac[1]], [d["gt"] for d in orac[0] + orac[1]], rpn_keys[0], [torch.zeros(2000)] * 4, False, mean, pstd)
        _, props_uns, _ = O.rcnn_student_losses(sd_s, [d["image"] for d in orac[2]], ps, rpn_keys[1], [torch.zeros(2000)] * 2, True, mean, pstd)
    keys = dict(rpn_sup=rpn_keys[0], rpn_unsup=rpn_keys[1],
                roi_sup=compact_roi(roi_keys[0], [len(p["boxes"]) for p in props_sup], [len(d["gt"]["boxes"]) for d in orac[0] + orac[1]]),
                roi_unsup=compact_roi(roi_keys[1], [len(p["boxes"]) for p in props_uns], [len(p["boxes"]) for p in ps]))
    rec_o, *_ = O.rcnn_semisup_step(sd_s, sd_t, orac, keys, keep_rate=cfg.SEMISUPNET.EMA_KEEP_RATE, lam_u=cfg.SEMISUPNET.UNSUP_LOSS_WEIGHT, lam_r=cfg.SEMISUPNET.UNSUP_REG_LOSS_WEIGHT,
                                    thr=cfg.SEMISUPNET.BBOX_THRESHOLD, lr=0.01, mean=mean, pix_std=pstd, pseudo_override=ps)
    print(name, {k: "%.2e" % (abs(rec[k] - v) / max(abs(v), 1e-12)) for k, v in rec_o.items() if k.startswith("loss")})
# anchor labelling of the unsup pass under both pseudo sets
from tests.test_rcnn_conditioning import _setup  # noqa
hw = [(-(-H // s), -(-W // s)) for s in (4, 8, 16, 32, 64)]
anchors = torch.cat(O.make_anchors(hw, [4, 8, 16, 32, 64]))
for i in range(2):
    a = O.matcher(O.pairwise_iou(pseudo[i]["boxes"], anchors), [0.3, 0.7], [0, -1, 1], True)
    b = O.matcher(O.pairwise_iou(prod_pseudo[i]["boxes"], anchors), [0.3, 0.7], [0, -1, 1], True)
    print("image", i, "labels differ at", int((a[1] != b[1]).sum()), "anchors; matched idx differ at", int((a[0] != b[0]).sum()), "; positives", int((a[1] == 1).sum()), int((b[1] == 1).sum()))
    iou = O.pairwise_iou(pseudo[i]["boxes"], anchors)
    print("  ties per pseudo box (oracle)", (iou == iou.max(dim=1)[0][:, None]).sum(dim=1).tolist())
torch.save({"oracle": pseudo, "product": prod_pseudo}, os.path.join(ROOT, "gpurun_out", "rcnn_coupled_pseudo.pt"))
for i in range(2):
    ia, ib = O.pairwise_iou(pseudo[i]["boxes"], anchors), O.pairwise_iou(prod_pseudo[i]["boxes"], anchors)
    a = O.matcher(ia, [0.3, 0.7], [0, -1, 1], True); b = O.matcher(ib, [0.3, 0.7], [0, -1, 1], True)
    for j in torch.nonzero(a[1] != b[1]).flatten().tolist():
        gi = int(a[0][j])
        # which gt made it a low-quality positive
        lqa = torch.nonzero(ia[:, j] == ia.max(dim=1)[0]).flatten().tolist(); lqb = torch.nonzero(ib[:, j] == ib.max(dim=1)[0]).flatten().tolist()
        print("img", i, "anchor", j, anchors[j].tolist(), "labels", int(a[1][j]), int(b[1][j]), "lowq gts", lqa, lqb)
        for gq in set(lqa + lqb):
            print("    gt", gq, "oracle box", pseudo[i]["boxes"][gq].tolist(), "product box", prod_pseudo[i]["boxes"][gq].tolist(),
                  "iou %.9f %.9f best %.9f %.9f  nties %d %d" % (float(ia[gq, j]), float(ib[gq, j]), float(ia[gq].max()), float(ib[gq].max()),
                                                           int((ia[gq] == ia[gq].max()).sum()), int((ib[gq] == ib[gq].max()).sum())))
