"""Eager step vs the step replayed as a hipGraph at a given per-GPU batch: host enqueue time and wall time per step.
usage: python tools/small_batch_probe.py [fcos|rcnn] [f16|bf16] [images per list] [steps] [small]
(env: UTV2_OVERLAP_TEACHER=0 UTV2_WGRAD_STREAM=0 -> the step on ONE stream, to see what the multi-stream topology costs a replay)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config

model = sys.argv[1] if len(sys.argv) > 1 else "fcos"
dtype = sys.argv[2] if len(sys.argv) > 2 else ("f16" if model == "fcos" else "bf16")
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
K = int(sys.argv[4]) if len(sys.argv) > 4 else 30
small = len(sys.argv) > 5 and sys.argv[5] == "small"
bench.set_amp_type(dtype)      # (the package's AMP default is fp16 since round 6: name the 16-bit type explicitly)
cfg = get_config(model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", B, "SOLVER.IMG_PER_BATCH_UNLABEL", B, "SEMISUPNET.BURN_UP_STEP", 0,
                            "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])


def make():
    torch.manual_seed(0)
    loader = None
    if small:
        from ubteacher.data.synthetic import SyntheticTwoCropLoader
        loader = SyntheticTwoCropLoader(cfg, height=96, width=128)
    tr = (UBRCNNTeacherTrainer if model == "rcnn" else UBTeacherTrainer)(cfg, data_loader=loader)
    tr.iter = 1; tr.log_period = 10 ** 9
    tr.optimizer.param_groups[0]["lr"] = 1e-12
    (bench.tune_rcnn_for_pseudo_labels if model == "rcnn" else bench.tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])
    return tr


def timed(step, n):
    from ubteacher.engine.step_gc import StepGC
    with StepGC() as g:          # UTV2_STEP_GC=0: the interpreter's default collector
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(); g.tick()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


tag = "%s %s %d+%d%s ovl=%s wgs=%s" % (model, dtype, B, B, " small" if small else "", os.environ.get("UTV2_OVERLAP_TEACHER", "1"), os.environ.get("UTV2_WGRAD_STREAM", "1"))
tr = make()


def eager():
    tr.run_step_full_semisup(); tr.iter += 1
for _ in range(6):
    eager()
e = timed(eager, K)
print("%s: eager  enqueue %.2f ms wall %.2f ms/step = %.1f img/s" % (tag, e[0], e[1], 2 * B / e[1] * 1e3), flush=True)
if os.environ.get("PROBE_NO_GRAPH") != "1":
    def graph():
        tr.run_step_graph(); tr.iter += 1
    try:
        for _ in range(5):
            graph()
        g = timed(graph, K)
        print("%s: graph  enqueue %.2f ms wall %.2f ms/step = %.1f img/s" % (tag, g[0], g[1], 2 * B / g[1] * 1e3), flush=True)
    except Exception as ex:  # noqa: BLE001
        print("%s: graph FAILED %r" % (tag, repr(ex)[:300]), flush=True)
