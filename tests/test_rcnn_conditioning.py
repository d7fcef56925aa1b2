"""Why the two pseudo-label RPN terms of the COUPLED Faster-RCNN step comparisons carry looser bounds than the north star's 1e-3
(tests/test_rcnn_step_gpu.py and bench.py parity_fullsize `looser_terms`: loss_rpn_loc_pseudo 2e-2 .. 6e-2, loss_rpn_cls_pseudo 5e-3) -
shown on the ORACLE alone (CPU, no product code): a property of the reference's algorithm, not of an implementation.

The reference labels RPN anchors against the teacher's pseudo boxes with Detectron2's Matcher, allow_low_quality_matches=True
(ubteacher/modeling/proposal_generator/rpn.py:112-148 [D2-recall]): every anchor whose IoU with a pseudo box EQUALS that box's best
IoU becomes a positive - `iou == best[:, None]`, exact floating-point equality.  For a pseudo box that CONTAINS anchors, all contained
anchors of one size have the same mathematical IoU, area(anchor) / area(box), whatever their position and aspect ratio (the three
ratios of a size share the area): a many-way tie.  Whether the fp32 values tie too depends on the last bit of each anchor's
`inter = w * h` (exact for the square anchors on integer coordinates, rounded for the 1:2 / 2:1 ones) against the rounding of
`area(box) + area(anchor) - inter`.  Move the box by a few ulp and a different subset of the contained anchors ties at the maximum.

The two boxes below are REAL: one pseudo box of tests/test_rcnn_step_gpu.py's problem as the oracle's teacher (CPU, fp32) and as the
product's teacher (MI355X, exact-f32 mode) computed it - they agree to 2.7e-5 px, as two correct fp32 implementations of a ResNet-50
+ FPN + RoIAlign + box head do (different accumulation orders).  Under the first, 18 anchors are low-quality positives, under the
second 8.  loss_rpn_loc is a SUM over the sampled positives (rpn.py:153-225): it moves by tens of per cent for this image; diluted over
a 256-anchor sample of a whole batch, by a few per cent.  Given the SAME pseudo boxes both terms agree to 1e-6
(tests/test_rcnn_step_gpu.py::test_rcnn_step_fp32_tight_with_the_product_pseudo_boxes; tools/debug_rcnn_coupled.py prints both)."""
import torch

from oracle import utv2_oracle as O

# fp32 values, exactly as produced (hex floats): pseudo box 25 of the second unlabeled image
BOX_ORACLE = [float.fromhex(h) for h in ("0x1.515dbap+5", "0x1.5cf794p+1", "0x1.7fe716p+6", "0x1.43f662p+5")]
BOX_PRODUCT = [float.fromhex(h) for h in ("0x1.515dc8p+5", "0x1.5cf786p+1", "0x1.7fe718p+6", "0x1.43f662p+5")]


def _anchors():
    hw = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]            # the FPN levels p2..p6 of a 96 x 128 image
    return torch.cat(O.make_anchors(hw, [4, 8, 16, 32, 64]))


def _rpn_terms(box, anchors, obj, dl, keys):
    out, samples = O.rpn_losses(anchors, obj, dl, [dict(boxes=torch.tensor([box], dtype=torch.float32), scores=torch.tensor([0.8551]))], keys, True)
    return {k: float(v) for k, v in out.items()}, len(samples[0][0])


def test_two_correct_fp32_teachers_select_different_low_quality_positives():
    anchors = _anchors()
    a, b = torch.tensor([BOX_ORACLE]), torch.tensor([BOX_PRODUCT])
    assert float((a - b).abs().max()) < 3e-5                                    # the two teachers agree to 2.7e-5 px
    ia, ib = O.pairwise_iou(a, anchors), O.pairwise_iou(b, anchors)
    assert abs(float(ia.max()) - float(ib.max())) < 3e-7                        # ... and so do the best IoUs
    ta, tb = int((ia == ia.max()).sum()), int((ib == ib.max()).sum())
    assert (ta, tb) == (18, 8), (ta, tb)                                        # but 18 anchors tie at the maximum under one, 8 under the other
    tied = torch.nonzero((ia == ia.max())[0]).flatten()
    wh = anchors[tied, 2:] - anchors[tied, :2]
    area = wh[:, 0] * wh[:, 1]
    assert float((area - 1024.0).abs().max()) < 1e-2                            # all of them 1024 px^2 anchors (32 x 32, 45 x 23, 23 x 45) ...
    inside = (anchors[tied, :2] >= a[0, :2]).all(dim=1) & (anchors[tied, 2:] <= a[0, 2:]).all(dim=1)
    assert bool(inside.all())                                                   # ... that lie INSIDE the pseudo box: IoU = area(anchor) / area(box)
    _, la = O.matcher(ia, [0.3, 0.7], [0, -1, 1], True)
    _, lb = O.matcher(ib, [0.3, 0.7], [0, -1, 1], True)
    assert int((la == 1).sum()) == 18 and int((lb == 1).sum()) == 8             # the positives of this image ARE the ties (best IoU 0.50 < 0.7)


def test_the_rpn_pseudo_terms_inherit_the_discontinuity():
    anchors = _anchors()
    R = anchors.shape[0]
    g = torch.Generator().manual_seed(3)
    obj = torch.randn(1, R, generator=g)
    dl = torch.randn(1, R, 4, generator=g) * 0.1
    keys = torch.rand(1, R, generator=g)
    la, na = _rpn_terms(BOX_ORACLE, anchors, obj, dl, keys)
    lb, nb = _rpn_terms(BOX_PRODUCT, anchors, obj, dl, keys)
    assert (na, nb) == (18, 8)
    rel = {k: abs(la[k] - lb[k]) / abs(la[k]) for k in la}
    assert rel["loss_rpn_loc"] > 0.1, rel            # a sum over 18 vs 8 positives: far above the north star's 1e-3
    assert rel["loss_rpn_cls"] > 1e-3, rel           # the BCE of the sampled anchors: ten of them change sides
    # the same two boxes through everything that is CONTINUOUS in them agree to fp32 accuracy: the size of the perturbation is 1e-7 relative
    tgt_a = O.rpn_get_deltas(anchors[:64], torch.tensor([BOX_ORACLE]).expand(64, 4))
    tgt_b = O.rpn_get_deltas(anchors[:64], torch.tensor([BOX_PRODUCT]).expand(64, 4))
    assert float((tgt_a - tgt_b).abs().max()) < 1e-5
