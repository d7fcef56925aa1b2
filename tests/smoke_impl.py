"""__graft_entry__.smoke(): one tiny UTv2 FCOS step on cuda:0 through the HIP path, checked against
the CPU oracle (losses within 1e-3 relative, same pseudo labels, teacher EMA bit exact)."""
import torch

from oracle import utv2_oracle as O
from tests.utv2_testutil import FixedLoader, cpu_state, make_batch, small_fcos_cfg, tune_state_for_pseudo_labels


def run():
    from ubteacher.engine import UBTeacherTrainer
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    import os
    if (os.cpu_count() or 1) > 16:
        torch.set_num_threads(16)      # the oracle leg: stock torch CPU kernels on small tensors slow down on the boxes' 128-256 threads (tests/conftest.py)
    H, W = 96, 128
    cfg = small_fcos_cfg(device="cuda:0")
    torch.manual_seed(0)
    prod, orac = make_batch(21, 2, 2, H, W, "cuda:0")
    tr = UBTeacherTrainer(cfg, data_loader=FixedLoader(prod))
    sd_s = tune_state_for_pseudo_labels(cpu_state(tr.model), [d["image"] for d in orac[3]])
    sd_t = dict(sd_s)
    sd_t["proposal_generator.fcos_head.bbox_pred_std.bias"] = torch.full((4,), -3.0)
    tr.model.load_state_dict(sd_s)
    tr.model_teacher.load_state_dict(sd_t)
    tr.iter = 1
    tr.optimizer.param_groups[0]["lr"] = 0.01
    tr.run_step_full_semisup()
    rec = tr.flush_metrics()
    rec_o, new_s, new_t, _, _, pseudo = O.fcos_semisup_step(
        O.FCOSCfg(), sd_s, sd_t, orac, keep_rate=cfg.SEMISUPNET.EMA_KEEP_RATE, lam_u=cfg.SEMISUPNET.UNSUP_LOSS_WEIGHT,
        lam_r=cfg.SEMISUPNET.UNSUP_REG_LOSS_WEIGHT, lr=0.01, mean=sd_s["pixel_mean"], pix_std=sd_s["pixel_std"])
    for k, v in rec_o.items():
        assert abs(rec[k] - v) <= 1e-3 * max(abs(v), 1e-6), (k, rec[k], v)
    t_after = cpu_state(tr.model_teacher)
    for k in new_t:
        assert torch.equal(t_after[k], new_t[k]), k
    print("smoke ok:", {k: round(v, 6) for k, v in rec.items() if k.startswith("loss")})
