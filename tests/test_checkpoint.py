"""Checkpoint surface (reference checkpoint/detection_checkpoint.py:10-89): a Caffe2 / Detectron ImageNet
R-50 pickle loads into the STUDENT only through Detectron2's blob-name conversion + longest-suffix matching; teacher/student
checkpoints round-trip with the reference's `modelTeacher.* / modelStudent.*` keys.  CPU only."""
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))


def _to_c2_name(k):
    """inverse of the D2 naming for the ResNet body (test-side generator of a synthetic R-50.pkl)"""
    k = k.replace("backbone.bottom_up.", "")
    if k.startswith("stem.conv1."):
        rest = k[len("stem.conv1."):]
        return {"weight": "conv1_w", "norm.weight": "res_conv1_bn_s", "norm.bias": "res_conv1_bn_b"}.get(rest)
    stage, blk, conv, rest = k.split(".", 3)
    br = {"shortcut": "branch1", "conv1": "branch2a", "conv2": "branch2b", "conv3": "branch2c"}[conv]
    suffix = {"weight": "w", "norm.weight": "bn_s", "norm.bias": "bn_b"}.get(rest)
    return None if suffix is None else "%s_%s_%s_%s" % (stage, blk, br, suffix)


def test_known_name_conversions():
    from ubteacher.checkpoint import convert_c2_detectron_names
    w = {"conv1_w": 0, "res_conv1_bn_s": 1, "res_conv1_bn_b": 2, "res2_0_branch1_w": 3, "res2_0_branch1_bn_s": 4,
         "res3_1_branch2a_w": 5, "res4_5_branch2b_bn_b": 6, "res5_2_branch2c_w": 7, "fc1000_w": 8, "fc1000_b": 9, "conv1_w_momentum": 10}
    got = convert_c2_detectron_names(w)
    assert got == {"stem.conv1.weight": 0, "stem.conv1.norm.weight": 1, "stem.conv1.norm.bias": 2, "res2.0.shortcut.weight": 3,
                   "res2.0.shortcut.norm.weight": 4, "res3.1.conv1.weight": 5, "res4.5.conv2.norm.bias": 6, "res5.2.conv3.weight": 7}


def test_c2_pickle_loads_into_student_only(tmp_path):
    from ubteacher.checkpoint import DetectionTSCheckpointer
    from ubteacher.modeling import build_model
    from ubteacher.modeling.ts_ensemble import EnsembleTSModel
    from ubteacher.presets import get_config
    cfg = get_config("fcos", 1, ["MODEL.DEVICE", "cpu"])
    torch.manual_seed(0)
    student, teacher = build_model(cfg), build_model(cfg)
    rng = np.random.default_rng(0)
    blobs, expect = {}, {}
    for k, v in student.state_dict().items():
        if not k.startswith("backbone.bottom_up."):
            continue
        c2 = _to_c2_name(k)
        if c2 is None:           # running_mean / running_var: absent from R-50.pkl (affine-only BN blobs)
            continue
        arr = rng.standard_normal(tuple(v.shape)).astype(np.float32)
        blobs[c2] = arr
        expect[k] = torch.from_numpy(arr)
    blobs["fc1000_w"] = rng.standard_normal((1000, 2048)).astype(np.float32)
    blobs["fc1000_b"] = np.zeros(1000, np.float32)
    blobs["conv1_w_momentum"] = np.zeros((64, 3, 7, 7), np.float32)
    path = os.path.join(tmp_path, "R-50.pkl")
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs}, f)
    t_before = {k: v.clone() for k, v in teacher.state_dict().items()}
    head_before = student.state_dict()["proposal_generator.fcos_head.cls_logits.weight"].clone()
    ens = EnsembleTSModel(teacher, student)
    ck = DetectionTSCheckpointer(ens, str(tmp_path))
    ck.load(path)
    sd = student.state_dict()
    assert len(expect) == 53 * 3                                  # 53 convs x {weight, norm.weight, norm.bias}
    for k, v in expect.items():
        assert torch.equal(sd[k], v), k                            # every body blob landed (NCHW view of the NHWC arena)
    assert torch.equal(sd["proposal_generator.fcos_head.cls_logits.weight"], head_before)   # nothing else touched
    assert ck.last_load_report["unmatched_checkpoint_keys"] == []   # fc1000 / momentum were dropped by the conversion
    for k, v in teacher.state_dict().items():                      # teacher untouched (student-only load)
        assert torch.equal(v, t_before[k]), k


def test_teacher_student_roundtrip(tmp_path):
    from ubteacher.checkpoint import DetectionTSCheckpointer
    from ubteacher.modeling import build_model
    from ubteacher.modeling.ts_ensemble import EnsembleTSModel
    from ubteacher.presets import get_config
    cfg = get_config("fcos", 1, ["MODEL.DEVICE", "cpu"])
    torch.manual_seed(1)
    a = EnsembleTSModel(build_model(cfg), build_model(cfg))
    torch.manual_seed(2)
    b = EnsembleTSModel(build_model(cfg), build_model(cfg))
    DetectionTSCheckpointer(a, str(tmp_path)).save("model_0000009", iteration=9)
    keys = list(torch.load(os.path.join(tmp_path, "model_0000009.pth"))["model"].keys())
    assert all(k.startswith("modelTeacher.") or k.startswith("modelStudent.") for k in keys)
    ckb = DetectionTSCheckpointer(b, str(tmp_path))
    assert ckb.has_checkpoint()
    out = ckb.resume_or_load("", resume=True)
    assert out["iteration"] == 9
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k


class _Opt:
    def __init__(self):
        self.mom = torch.zeros(4)
        self.lr = 0.1

    def state_dict(self):
        return {"momentum_buffer": self.mom, "lr": self.lr}

    def load_state_dict(self, sd):
        if "momentum_buffer" not in sd:
            raise ValueError("not an ArenaSGD state")
        self.mom = sd["momentum_buffer"].clone()
        self.lr = sd["lr"]


class _Sched:
    def __init__(self):
        self.last_iter = 0

    def state_dict(self):
        return {"last_iter": self.last_iter}

    def load_state_dict(self, sd):
        self.last_iter = sd["last_iter"]


def test_weights_path_resolution_and_non_resume_semantics(tmp_path, monkeypatch):
    """MODEL.WEIGHTS handling (Detectron2 Checkpointer.resume_or_load + PathManager [D2-recall]): an unresolvable path raises instead of
    silently training from random weights; detectron2:// URIs map into the local model-zoo cache; initialising from a checkpoint
    WITHOUT resuming takes the model only (no optimizer / scheduler / iteration); a foreign optimizer state is refused on resume."""
    import pytest
    from ubteacher.checkpoint import DetectionCheckpointer, DetectionTSCheckpointer, resolve_path
    from ubteacher.modeling import build_model
    from ubteacher.modeling.ts_ensemble import EnsembleTSModel
    from ubteacher.presets import get_config
    cfg = get_config("fcos", 1, ["MODEL.DEVICE", "cpu"])
    torch.manual_seed(3)
    a = EnsembleTSModel(build_model(cfg), build_model(cfg))
    oa, sa = _Opt(), _Sched()
    oa.mom += 5.0; oa.lr = 0.01; sa.last_iter = 77
    src = tmp_path / "src"
    DetectionTSCheckpointer(a, str(src), optimizer=oa, scheduler=sa).save("model_0000076", iteration=76)
    weights = str(src / "model_0000076.pth")

    torch.manual_seed(4)
    b = EnsembleTSModel(build_model(cfg), build_model(cfg))
    ob, sb = _Opt(), _Sched()
    ck = DetectionTSCheckpointer(b, str(tmp_path / "fresh"), optimizer=ob, scheduler=sb)
    with pytest.raises(FileNotFoundError):
        ck.resume_or_load(str(tmp_path / "nope.pth"), resume=False)
    with pytest.raises(FileNotFoundError):
        ck.resume_or_load("detectron2://ImageNetPretrained/MSRA/R-50.pkl", resume=True)   # the shipped configs' default: not cached here
    assert ck.resume_or_load("", resume=False) == {}                                        # explicit "from scratch"
    out = ck.resume_or_load(weights, resume=False)
    assert "iteration" not in out and "optimizer" not in out
    assert float(ob.mom.abs().sum()) == 0.0 and ob.lr == 0.1 and sb.last_iter == 0        # checkpointables=[]
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    # resume: everything comes back
    ck2 = DetectionTSCheckpointer(b, str(src), optimizer=ob, scheduler=sb)
    out = ck2.resume_or_load("", resume=True)
    assert out["iteration"] == 76 and ob.lr == 0.01 and sb.last_iter == 77 and float(ob.mom[0]) == 5.0
    # a reference-produced checkpoint carries a torch.optim state_dict: resuming takes weights / scheduler / iteration and warns that the
    # momentum restarts (strict_optimizer = True refuses with a clear message instead)
    data = torch.load(weights)
    data["optimizer"] = {"state": {0: {"momentum_buffer": torch.zeros(3)}}, "param_groups": [{"lr": 0.01}]}
    torch.save(data, weights)
    ck2.strict_optimizer = True
    with pytest.raises(ValueError, match="ArenaSGD"):
        ck2.resume_or_load("", resume=True)
    ck2.strict_optimizer = False
    sb.last_iter = 0
    out = ck2.resume_or_load("", resume=True)
    assert out["iteration"] == 76 and sb.last_iter == 77 and ck2.last_optimizer_skipped
    ck.resume_or_load(weights, resume=False)                                                # ... and simply skipped otherwise
    # model-zoo cache mapping
    monkeypatch.setenv("FVCORE_CACHE", str(tmp_path / "cache"))
    zoo = tmp_path / "cache" / "detectron2" / "ImageNetPretrained" / "MSRA"
    zoo.mkdir(parents=True)
    (zoo / "R-50.pkl").write_bytes(b"x")
    assert resolve_path("detectron2://ImageNetPretrained/MSRA/R-50.pkl") == str(zoo / "R-50.pkl")
    # plain single-model checkpointer (Faster-RCNN --eval-only): refuses a teacher/student checkpoint instead of loading nothing
    torch.manual_seed(5)
    m = build_model(cfg)
    with pytest.raises(ValueError, match="modelTeacher"):
        DetectionCheckpointer(m, str(tmp_path / "e")).resume_or_load(weights, resume=False)
    plain = tmp_path / "plain.pth"
    torch.save({"model": {k: v.clone() for k, v in a.modelTeacher.state_dict().items()}}, str(plain))
    DetectionCheckpointer(m, str(tmp_path / "e")).resume_or_load(str(plain), resume=False)
    for (k, v), (_, w) in zip(a.modelTeacher.state_dict().items(), m.state_dict().items()):
        assert torch.equal(v, w), k


def test_arena_sgd_state_is_per_key_and_layout_independent(tmp_path, monkeypatch):
    """the momentum arena's layout follows the parameter arena's, which depends on the build (paired FCOS towers merge
    cls_tower / bbox_tower tensors into one handle).  The optimizer state is saved per state_dict key through the weights' export views:
    a checkpoint written under one layout resumes under another with every tensor's momentum on that tensor; a torch.optim.SGD
    state (the reference's checkpoints) maps by Detectron2's parameter order with shape checks; the old flat form is refused."""
    import pytest
    from ubteacher.engine.trainer import ArenaSGD
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    cfg = get_config("fcos", 1, ["MODEL.DEVICE", "cpu"])
    torch.manual_seed(0)
    a = build_model(cfg)
    oa = ArenaSGD(cfg, a)
    g = torch.Generator().manual_seed(1)
    oa.store.mom.copy_(torch.randn(oa.store.mom.shape, generator=g))
    oa.param_groups[0]["lr"] = 0.0123
    sd = oa.state_dict()
    assert "momentum_buffer" not in sd and list(sd["momentum"]) == [k for k in a.state_dict() if k in sd["momentum"]]
    k_cls, k_box = "proposal_generator.fcos_head.cls_tower.3.weight", "proposal_generator.fcos_head.bbox_tower.3.weight"
    assert tuple(sd["momentum"][k_cls].shape) == tuple(a.state_dict()[k_cls].shape) == (256, 256, 3, 3)
    assert not torch.equal(sd["momentum"][k_cls], sd["momentum"][k_box])
    # another layout of the same model: the two towers as separate chains
    monkeypatch.setenv("UTV2_PAIR_TOWERS", "0")
    b = build_model(cfg)
    monkeypatch.delenv("UTV2_PAIR_TOWERS")
    ob = ArenaSGD(cfg, b)
    assert [h.shape for h in a.store.handles] != [h.shape for h in b.store.handles]          # the arenas really differ
    assert list(a.state_dict()) == list(b.state_dict())                                       # the surface does not
    ob.load_state_dict(sd)
    assert ob.param_groups[0]["lr"] == 0.0123
    back = ob.state_dict()["momentum"]
    for k, v in sd["momentum"].items():
        assert torch.equal(back[k], v), k
    assert not torch.equal(oa.store.mom, ob.store.mom)            # a flat copy would have been wrong
    # through a file
    from ubteacher.checkpoint import DetectionTSCheckpointer
    from ubteacher.modeling.ts_ensemble import EnsembleTSModel
    DetectionTSCheckpointer(EnsembleTSModel(a, a), str(tmp_path), optimizer=oa).save("model_0000001", iteration=1)
    ob.store.mom.zero_()
    ck = DetectionTSCheckpointer(EnsembleTSModel(b, b), str(tmp_path), optimizer=ob)
    ck.resume_or_load("", resume=True)
    assert not ck.last_optimizer_skipped
    for k, v in sd["momentum"].items():
        assert torch.equal(ob.state_dict()["momentum"][k], v), k
    # the flat form of rounds 1-3 carries no layout: refused
    with pytest.raises(ValueError, match="flat"):
        ob.load_state_dict({"momentum_buffer": oa.store.mom.clone(), "lr": 0.1})
    # a torch.optim.SGD state in Detectron2's parameter order: weights / biases (group 0), then norm parameters (group 1)
    keys = list(sd["momentum"])
    norm = [k for k in keys if ".cls_tower." in k or ".bbox_tower." in k]
    norm = [k for k in norm if int(k.split("_tower.")[1].split(".")[0]) % 3 == 1]            # GroupNorm weight / bias
    assert len(norm) == 16
    dec = [k for k in keys if k not in norm]
    state = {i: {"momentum_buffer": sd["momentum"][k].clone()} for i, k in enumerate(dec + norm)}
    torch_sd = {"state": state, "param_groups": [{"lr": 0.5, "weight_decay": 1e-4, "params": list(range(len(dec)))},
                                                 {"lr": 0.5, "weight_decay": 0.0, "params": list(range(len(dec), len(keys)))}]}
    ob.store.mom.zero_()
    ob.load_state_dict(torch_sd)
    assert ob.param_groups[0]["lr"] == 0.5
    for k, v in sd["momentum"].items():
        assert torch.equal(ob.state_dict()["momentum"][k], v), k
    # older Detectron2 releases (no reduce_param_groups) emit ONE GROUP PER PARAMETER in named_parameters() order - conv and
    # GroupNorm tensors interleaved, [256] biases next to [256] norm weights; the ids then follow the model's key order
    per_param = {"state": {i: {"momentum_buffer": sd["momentum"][k].clone()} for i, k in enumerate(keys)},
                 "param_groups": [{"lr": 0.25, "weight_decay": 0.0 if k in norm else 1e-4, "params": [i]} for i, k in enumerate(keys)]}
    ob.store.mom.zero_()
    ob.load_state_dict(per_param)
    assert ob.param_groups[0]["lr"] == 0.25
    for k, v in sd["momentum"].items():
        assert torch.equal(ob.state_dict()["momentum"][k], v), k
    # grouped layouts that cannot be verified are refused instead of shape-matched: equal weight decay in both groups, wrong group sizes
    before = ob.store.mom.clone()
    same_wd = {"state": state, "param_groups": [dict(g, weight_decay=1e-4) for g in torch_sd["param_groups"]]}
    with pytest.raises(ValueError, match="cannot be told apart"):
        ob.load_state_dict(same_wd)
    shifted = {"state": state, "param_groups": [dict(torch_sd["param_groups"][0], params=list(range(len(dec) - 1))),
                                                 dict(torch_sd["param_groups"][1], params=list(range(len(dec) - 1, len(keys))))]}
    with pytest.raises(ValueError, match="cannot be told apart"):
        ob.load_state_dict(shifted)
    assert torch.equal(ob.store.mom, before)
    state[3]["momentum_buffer"] = torch.zeros(7)                                              # a shape that cannot be this tensor's
    with pytest.raises(ValueError, match="size mismatch"):
        ob.load_state_dict(torch_sd)
    assert torch.equal(ob.store.mom, before)                                                  # nothing written on failure


def test_amp_scaler_state_round_trips(tmp_path):
    """the fp16 loss-scale state (scale, growth tracker) is a checkpointable (`grad_scaler`), not restarted at 65536"""
    from ubteacher.checkpoint import DetectionTSCheckpointer
    from ubteacher.engine.trainer import AmpScalerState

    class T:
        _amp_state = torch.tensor([1024.0, 0.0, 37.0])
    t1, t2 = T(), T()
    t2._amp_state = torch.tensor([65536.0, 0.0, 0.0])

    class M:
        def state_dict(self):
            return {"modelStudent.w": torch.zeros(1)}

        def load_state_dict(self, sd, strict=False):
            pass
    w = DetectionTSCheckpointer(M(), str(tmp_path), grad_scaler=AmpScalerState(t1))
    w.save("model_0000005", iteration=5)
    w.save("model_final", iteration=9)
    # the checkpoint files carry the CALLER's names (an extra checkpointable's key used to shadow the `name` argument:
    # every save went to grad_scaler.pth and model_final.pth never existed)
    import os
    assert sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".pth")) == ["model_0000005.pth", "model_final.pth"]
    assert open(os.path.join(str(tmp_path), "last_checkpoint")).read().strip() == "model_final.pth"
    ck = DetectionTSCheckpointer(M(), str(tmp_path), grad_scaler=AmpScalerState(t2))
    assert ck.last_optimizer_skipped is False            # initialised (the attribute used to exist only after a skipped load)
    ck.resume_or_load("", resume=True)
    assert t2._amp_state.tolist() == [1024.0, 0.0, 37.0]
    t3 = T(); t3._amp_state = None                       # bf16 / f32 run resuming an fp16 checkpoint: ignored
    DetectionTSCheckpointer(M(), str(tmp_path), grad_scaler=AmpScalerState(t3)).resume_or_load("", resume=True)


def test_roi_heads_loss_selects_the_predictor_and_its_state_dict_surface():
    """MODEL.ROI_HEADS.LOSS (reference roi_heads/roi_heads.py:52-66): "FocalLoss" = the UTv1 predictor on Detectron2's FastRCNNOutputLayers
    (cls_score [K+1], bbox_pred [4K], no bbox_pred_std); the *_BoundaryVar predictors add the class-agnostic bbox_pred / bbox_pred_std;
    "CrossEntropy" cannot train in the reference either (its ROI heads call Detectron2's two-argument losses() with three) and says so."""
    import pytest
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    shapes = {}
    S1 = ["MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE", "smooth_l1", "MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG", False]   # the UTv2 YAML: nlloss, agnostic
    for loss in ("FocalLoss", "FocalLoss_BoundaryVar", "CrossEntropy_BoundaryVar"):
        m = build_model(get_config("rcnn", 1, ["MODEL.DEVICE", "cpu", "MODEL.ROI_HEADS.LOSS", loss] + (S1 if loss == "FocalLoss" else [])))
        shapes[loss] = {k[len("roi_heads.box_predictor."):]: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("roi_heads.box_predictor.")}
    assert shapes["FocalLoss"] == {"cls_score.weight": (81, 1024), "cls_score.bias": (81,), "bbox_pred.weight": (320, 1024), "bbox_pred.bias": (320,)}
    want = {"cls_score.weight": (81, 1024), "cls_score.bias": (81,), "bbox_pred.weight": (4, 1024), "bbox_pred.bias": (4,),
            "bbox_pred_std.weight": (4, 1024), "bbox_pred_std.bias": (4,)}
    assert shapes["FocalLoss_BoundaryVar"] == want and shapes["CrossEntropy_BoundaryVar"] == want
    m = build_model(get_config("rcnn", 1, ["MODEL.DEVICE", "cpu", "MODEL.ROI_HEADS.LOSS", "FocalLoss"] + S1[:2]))
    assert tuple(m.state_dict()["roi_heads.box_predictor.bbox_pred.weight"].shape) == (4, 1024)
    w = m.state_dict()["roi_heads.box_predictor.cls_score.weight"]
    assert 0.005 < float(w.std()) < 0.02 and float(m.state_dict()["roi_heads.box_predictor.bbox_pred.weight"].std()) < 0.002    # D2: normal 0.01 / 0.001
    with pytest.raises(ValueError, match="Invalid bbox reg loss type 'nlloss'"):      # the UTv2 YAML's box loss on the UTv1 predictor (fast_rcnn.py:184-186)
        build_model(get_config("rcnn", 1, ["MODEL.DEVICE", "cpu", "MODEL.ROI_HEADS.LOSS", "FocalLoss"]))
    with pytest.raises(NotImplementedError, match="roi_heads.py:124"):
        build_model(get_config("rcnn", 1, ["MODEL.DEVICE", "cpu", "MODEL.ROI_HEADS.LOSS", "CrossEntropy"]))
    with pytest.raises(ValueError, match="Unknown ROI head loss"):
        build_model(get_config("rcnn", 1, ["MODEL.DEVICE", "cpu", "MODEL.ROI_HEADS.LOSS", "nope"]))


def test_weight_mirror_freshness_rules(monkeypatch):
    """ParamStore.mirror16 (the 16-bit copy of the arena the mixed-precision convs read, handed to the SGD / EMA kernels that write it
    themselves): a RANGE update (SGD) may only take it when it is fresh - the untouched ranges must already be valid -, a whole-arena
    update (EMA) whenever it exists with the selected library's element type; any other writer (load_state_dict -> touch()) leaves it stale
    for the lazy conversion; UTV2_FUSED_MIRROR=0 switches the hand-over off."""
    import torch
    from ubteacher import hip
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    m = build_model(get_config("fcos", 1, ["MODEL.DEVICE", "cpu"]))
    st = m.store
    assert st.mirror16(need_fresh=True) is None and st.mirror16(need_fresh=False) is None      # no mirror yet
    st._flat16 = torch.zeros(st.total, dtype=hip.h16_dtype())                                    # as bf16() creates it
    st._v16 = st.version - 1
    assert st.mirror16(need_fresh=True) is None and st.mirror16(need_fresh=False) is st._flat16  # stale: EMA may take it, SGD may not
    st._v16 = st.version
    assert st.mirror16(need_fresh=True) is st._flat16
    st.touch()                                                                                   # the update ran
    assert st._v16 != st.version
    st.mirror16_written()
    assert st._v16 == st.version and st.mirror16(need_fresh=True) is st._flat16
    m.load_state_dict(m.state_dict())                                                            # another writer: stale again
    assert st.mirror16(need_fresh=True) is None
    st._v16 = st.version
    monkeypatch.setenv("UTV2_FUSED_MIRROR", "0")
    assert st.mirror16(need_fresh=True) is None and st.mirror16(need_fresh=False) is None
    monkeypatch.delenv("UTV2_FUSED_MIRROR")
    st._flat16 = st._flat16.to(torch.float16 if hip.h16_dtype() == torch.bfloat16 else torch.bfloat16)   # the other library's type
    assert st.mirror16(need_fresh=False) is None
