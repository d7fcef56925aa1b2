"""A stand-in for R-50.pkl where none can be downloaded: backbone-only weights (student keys, loaded like the pretrained pickle:
DetectionTSCheckpointer -> student only) whose FrozenBN multipliers keep a RANDOM ResNet-50's activations at unit scale - the stem's
norm.weight absorbs the raw pixel scale (Faster-RCNN: pixel_std 1, values of +-100), every bottleneck's last norm.weight damps its
residual branch.  Random features, sane magnitudes: what a numerics soak of the recipes needs (a random backbone under FrozenBN's identity
statistics reaches 1e4-1e5 at the FPN and the Faster-RCNN recipe diverges from it within a few iterations).
usage: python tools/make_synthetic_backbone.py fcos|rcnn OUT.pth [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch  # noqa: E402

from ubteacher.modeling import build_model  # noqa: E402
from ubteacher.presets import get_config  # noqa: E402


def main():
    family, out = sys.argv[1], sys.argv[2]
    torch.manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    cfg = get_config(family, 1, ["MODEL.DEVICE", "cpu"])
    m = build_model(cfg)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith("backbone.bottom_up.")}
    pixel_scale = 60.0 / float(sum(cfg.MODEL.PIXEL_STD) / 3.0)      # magnitude of the normalised input image
    for k in sd:
        if k.endswith("stem.conv1.norm.weight"):
            sd[k].fill_(1.0 / pixel_scale)
        elif k.endswith("conv3.norm.weight"):
            sd[k].fill_(0.25)
        elif k.endswith("shortcut.norm.weight"):
            sd[k].fill_(0.7)
    torch.save({"model": sd}, out)
    print("wrote", out, len(sd), "tensors")


if __name__ == "__main__":
    main()
