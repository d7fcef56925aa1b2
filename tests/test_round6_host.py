"""Host-side pieces added in round 6 (no GPU): the training loop's collector policy, the ragged-canvas size draw of the synthetic loader,
the caller-owned table of recorded split-K tails, the AMP-type selection rule."""
import ctypes
import gc
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))


def test_step_gc_freezes_disables_and_restores(monkeypatch):
    from ubteacher.engine.step_gc import StepGC
    monkeypatch.delenv("UTV2_STEP_GC", raising=False)
    assert gc.isenabled()
    with StepGC(period=3) as g:
        assert not gc.isenabled() and gc.get_freeze_count() > 0
        calls = []
        monkeypatch.setattr(gc, "collect", lambda gen=2: calls.append(gen) or 0)
        for _ in range(7):
            g.tick()
        assert calls == [1, 1]                      # young generations, every `period` iterations, between two steps
    assert gc.isenabled() and gc.get_freeze_count() == 0
    monkeypatch.setenv("UTV2_STEP_GC", "0")           # the interpreter's default policy
    with StepGC() as g:
        assert gc.isenabled()
        g.tick()
    assert gc.isenabled()


def test_resize_shortest_edge_size_follows_detectron2s_rule():
    """ResizeShortestEdge with MIN_SIZE_TRAIN_SAMPLING "range" (the UTv2 recipes: (400, 1200), MAX_SIZE_TRAIN 1333): the short side is drawn
    from the range, the long side follows the aspect ratio and is capped - then both sides shrink together"""
    from ubteacher.data.synthetic import resize_shortest_edge_size
    rng = np.random.default_rng(0)
    seen_cap = seen_free = False
    for _ in range(500):
        h, w = resize_shortest_edge_size(rng, (400, 1200), 1333)
        assert w <= 1333 and 300 <= h <= 1200 and w >= h
        assert abs(w / h - 4.0 / 3.0) < 0.01
        if w == 1333:
            seen_cap = True
            assert h <= 1000
        else:
            seen_free = True
            assert 400 <= h <= 1200
    assert seen_cap and seen_free


def test_split_k_tail_table_is_caller_owned_host_memory():
    """utv2_conv2d_wgrad_bf16_d records a launch's tail in a table the CALLER owns (include/utv2.h): its size comes from the library, a
    zeroed table holds nothing, flushing nothing launches nothing (works without a GPU), a null table is an argument error"""
    from ubteacher import hip
    lib = hip.load()
    n = int(lib.utv2_wgrad_fold_table_bytes())
    assert 256 <= n <= 4096                              # passed by value as the flush kernel's argument
    buf = ctypes.create_string_buffer(n)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.utv2_wgrad_fold_pending(p) == 0
    assert lib.utv2_wgrad_fold_flush(p, ctypes.c_void_p(0)) == 0
    assert lib.utv2_wgrad_fold_pending(ctypes.c_void_p(0)) == -1
    assert lib.utv2_wgrad_fold_flush(ctypes.c_void_p(0), ctypes.c_void_p(0)) != 0
    for name in ("utv2_rowinfo_nhwc", "utv2_conv2d_wgrad_bf16_d", "utv2_wgrad_fold_flush"):
        assert name in hip.symbols()
    # rowinfo: argument checks happen on the host, before any launch
    assert lib.utv2_rowinfo_nhwc(ctypes.c_void_p(0), 1, 8, 8, 8, 8, 1, 1, 3, 3, 0, ctypes.c_void_p(0)) != 0      # no output
    assert lib.utv2_rowinfo_nhwc(ctypes.c_void_p(16), 1, 8, 8, 8, 8, 1, 1, 5, 5, 0, ctypes.c_void_p(0)) != 0     # 25 taps > 16


@pytest.mark.parametrize("env,amp,want", [(None, True, "fp16"), ("bf16", True, "bf16"), ("fp32", True, "fp32"), ("f16", True, "fp16"),
                                          (None, False, "fp32"), ("bf16", False, "fp32"), ("fp16", False, "fp32")])
def test_amp_type_selection_rule(env, amp, want, monkeypatch):
    """SOLVER.AMP.ENABLED (the reference's key, engine/trainer.py:194-198) selects the reference's own autocast type; UTV2_PRECISION names
    the 16-bit type of that path only (bf16 = opt-in, fp32 = back to exact fp32) and never turns a non-AMP config into a 16-bit one"""
    import torch
    from ubteacher import ops
    from ubteacher.engine import trainer as T
    if env is None:
        monkeypatch.delenv("UTV2_PRECISION", raising=False)
    else:
        monkeypatch.setenv("UTV2_PRECISION", env)

    class Store:
        flat = torch.zeros(1)

    class Stub(T._TrainerBase):
        def __init__(self):
            self.model = type("M", (), {"store": Store()})()

    class Cfg:
        class SOLVER:
            class AMP:
                ENABLED = amp
            MAX_ITER = 1
    st = Stub()
    # run only the precision rule of _common_init (the rest needs models): it is its first statement block
    prev = ops.PRECISION[0]
    try:
        try:
            T._TrainerBase._common_init(st, Cfg)
        except AttributeError:
            pass                                       # the stub has no data loader / checkpointer: the rule has run by then
        assert ops.PRECISION[0] == want
        assert (st._amp_state is not None) == (want == "fp16")
    finally:
        ops.set_precision(prev)


def test_unknown_amp_type_is_refused(monkeypatch):
    from ubteacher.engine import trainer as T
    monkeypatch.setenv("UTV2_PRECISION", "fp8")

    class Cfg:
        class SOLVER:
            class AMP:
                ENABLED = True
    with pytest.raises(ValueError):
        T._TrainerBase._common_init(type("S", (T._TrainerBase,), {"__init__": lambda self: None})(), Cfg)


@pytest.mark.parametrize("paired", ["1", "0"])
def test_tower_layers_are_chained_for_the_fused_groupnorm_backward(paired, monkeypatch):
    """ops.chain_gn_conv: inside a tower, layer i's GroupNorm output feeds layer i + 1's conv and nothing else - the link the
    fused GroupNorm backward rides on.  The LAST GroupNorm of a tower has no link (its gradient comes from the prediction convs' dgrads),
    the first conv has no source (it reads the FPN features); the switch defaults to on and UTV2_GN_BWD_FUSE=0 turns it off."""
    from ubteacher import ops
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    monkeypatch.setenv("UTV2_PAIR_TOWERS", paired)
    m = build_model(get_config("fcos", 1, ["MODEL.DEVICE", "cpu"]))
    head = m.proposal_generator.fcos_head
    towers = [head.towers["pair"]] if paired == "1" else [head.towers["cls"], head.towers["bbox"]]
    assert all(len(t) == 4 for t in towers)
    for layers in towers:
        assert layers[0][0].gnb_src is None and layers[-1][1].next_conv is None
        for i in range(1, len(layers)):
            assert layers[i][0].gnb_src is layers[i - 1][1] and layers[i - 1][1].next_conv is layers[i][0]
    assert head.cls_logits.gnb_src is None and head.box_head.gnb_src is None
    monkeypatch.delenv("UTV2_GN_BWD_FUSE", raising=False)
    assert ops.gn_bwd_fuse_on()
    monkeypatch.setenv("UTV2_GN_BWD_FUSE", "0")
    assert not ops.gn_bwd_fuse_on()
