"""The periodic writers and the evaluation cadence of the training loop (reference engine/trainer.py:503-552: build_hooks - LRScheduler,
PeriodicCheckpointer, two EvalHooks (student under `<key>_student`, then the teacher), PeriodicWriter(build_writers(), period=20)).
CPU: the storage / writer contract and the loop's cadence on a trainer whose step is a stub."""
import json
import logging
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
sys.path.insert(0, ROOT)


def test_event_storage_smoothing_hint_and_json_writer(tmp_path):
    from ubteacher.d2.events import EventStorage, JSONWriter
    path = tmp_path / "out" / "metrics.json"
    with EventStorage(0) as st:
        w = JSONWriter(str(path), window_size=3)
        for i in range(5):
            st.put_scalar("total_loss", float(i))                      # smoothed: median of the last 3
            st.put_scalar("lr", 0.1 * (i + 1), smoothing_hint=False)    # not smoothed: latest
            if i == 2:
                st.put_scalars(smoothing_hint=False, **{"bbox/AP": 12.5})
            if i in (2, 4):
                w.write(st)
            st.step()
        w.write(st)            # nothing new: no line
        w.close()
    lines = [json.loads(l) for l in path.read_text().splitlines()]
    assert [l["iteration"] for l in lines] == [2, 4]
    assert lines[0] == {"iteration": 2, "total_loss": 1.0, "lr": pytest.approx(0.3), "bbox/AP": 12.5}
    assert lines[1] == {"iteration": 4, "total_loss": 3.0, "lr": pytest.approx(0.5)}      # bbox/AP was written at its own iteration only
    # appended, not truncated, on re-open (resume)
    with EventStorage(5) as st:
        w = JSONWriter(str(path))
        st.put_scalar("total_loss", 9.0)
        w.write(st)
        w.close()
    assert len(path.read_text().splitlines()) == 3


def test_common_metric_printer_line(caplog):
    from ubteacher.d2.events import CommonMetricPrinter, EventStorage
    with EventStorage(19) as st, caplog.at_level(logging.INFO, logger="ubteacher.events"):
        st.put_scalars(total_loss=2.5, loss_fcos_cls=1.5, loss_fcos_loc=1.0, EMA_rate=0.9996, data_time=0.002)
        st.put_scalar("time", 0.025, smoothing_hint=False)
        st.put_scalar("lr", 0.01, smoothing_hint=False)
        st.put_scalars(smoothing_hint=False, **{"bbox/AP": 1.0})
        CommonMetricPrinter(max_iter=100, window_size=1).write(st)
    line = caplog.records[-1].getMessage()
    assert "iter: 19" in line and "total_loss: 2.5" in line and "loss_fcos_cls: 1.5" in line and "EMA_rate: 0.9996" in line
    assert "time: 0.0250" in line and "data_time: 0.0020" in line and "lr: 0.01" in line and "eta: 0:00:02" in line
    assert "bbox/AP" not in line          # evaluation results go to metrics.json, not to the console line


class _Sched:
    def __init__(self, opt):
        self.opt, self.n = opt, 0

    def step(self):
        self.n += 1
        self.opt.param_groups[0]["lr"] = 0.01 / (1 + self.n)


def _stub_trainer(tmp_path, max_iter, eval_period, ckpt_period):
    """a trainer whose step only produces metrics: the loop around it is the product's"""
    from ubteacher.config import add_ubteacher_config
    from ubteacher.d2 import get_cfg
    from ubteacher.engine.trainer import _TrainerBase
    cfg = get_cfg()
    add_ubteacher_config(cfg)
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.TEST.EVAL_PERIOD = eval_period
    cfg.SOLVER.CHECKPOINT_PERIOD = ckpt_period
    cfg.SOLVER.MAX_ITER = max_iter
    events = []

    class T(_TrainerBase):
        log_period = 4

        def __init__(self):
            self.cfg = cfg
            self.max_iter = max_iter
            self.storage = None
            self._pending_metrics = None
            self._last_metrics = {}

            class Opt:
                param_groups = [{"lr": 0.01}]
            self.optimizer = Opt()
            self.scheduler = _Sched(self.optimizer)

            class Ck:
                def save(self_, name, **kw):
                    events.append(("save", name, kw.get("iteration")))
            self.checkpointer = Ck()

            class M:
                pass
            self.model, self.model_teacher = M(), M()

        def run_step_full_semisup(self):
            self._pending_metrics = (["loss_a"], None, [("loss_a", 1.0 + self.iter), ("data_time", 0.001)])
            if (self.iter + 1) % self.log_period == 0:
                self.flush_metrics()

        def flush_metrics(self):
            if self._pending_metrics is None:
                return self._last_metrics
            _, _, host = self._pending_metrics
            self._pending_metrics = None
            md = dict(host)
            self.storage.put_scalar("data_time", md.pop("data_time"))
            self.storage.put_scalar("total_loss", sum(md.values()))
            self.storage.put_scalars(**md)
            self._last_metrics = md
            return md

        @classmethod
        def test(cls, cfg, model, evaluators=None):
            events.append(("test", "student" if model is tr.model else "teacher", tr.iter))
            return {"bbox": {"AP": 10.0 if model is tr.model else 20.0, "AP50": float("nan")}, "_speed": {"images": 3}}

    tr = T()
    return tr, events


def test_train_loop_writes_metrics_evaluates_and_checkpoints_on_the_reference_cadence(tmp_path):
    tr, events = _stub_trainer(tmp_path, max_iter=10, eval_period=6, ckpt_period=5)
    tr.train_loop(0, 10)
    # PeriodicCheckpointer: every CHECKPOINT_PERIOD iterations + model_final
    assert [e[1:] for e in events if e[0] == "save"] == [("model_0000004", 4), ("model_0000009", 9), ("model_final", 9)]
    # EvalHook x 2: student then teacher, after iteration 5 (6th) and once after the last iteration
    assert [e[1:] for e in events if e[0] == "test"] == [("student", 5), ("teacher", 5), ("student", 9), ("teacher", 9)]
    lines = [json.loads(l) for l in (tmp_path / "metrics.json").read_text().splitlines()]
    by_it = {l["iteration"]: l for l in lines}
    # PeriodicWriter(period = log_period = 4): iterations 3, 7 and the last one (9); evaluation scalars at the iteration they were made
    assert sorted(by_it) == [3, 5, 7, 9]
    assert by_it[3]["total_loss"] == 4.0 and by_it[3]["loss_a"] == 4.0 and by_it[7]["total_loss"] == 8.0 and by_it[9]["loss_a"] == 10.0
    assert by_it[3]["lr"] == pytest.approx(0.01 / 4) and by_it[9]["lr"] == pytest.approx(0.01 / 10)     # the rate the written step ran at
    assert "time" in by_it[3] and by_it[3]["time"] >= 0
    assert by_it[5] == {"iteration": 5, "bbox_student/AP": 10.0, "bbox/AP": 20.0}                     # NaN entries are dropped
    assert by_it[9]["bbox_student/AP"] == 10.0 and by_it[9]["bbox/AP"] == 20.0
    assert tr._last_eval_results_student["bbox"]["AP"] == 10.0 and tr._last_eval_results_teacher["bbox"]["AP"] == 20.0


def test_train_loop_without_evaluation_or_output_dir(tmp_path):
    tr, events = _stub_trainer(tmp_path, max_iter=5, eval_period=0, ckpt_period=0)
    tr.cfg.OUTPUT_DIR = ""
    tr.train_loop(2, 5)       # a resumed run: iterations 2, 3, 4
    assert not [e for e in events if e[0] == "test"] and not [e for e in events if e[0] == "save"]
    assert not (tmp_path / "metrics.json").exists()
