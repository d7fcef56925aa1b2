import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd")); sys.path.insert(0, ROOT)
import torch, ctypes
from ubteacher import hip, ops
from tests.test_gn_bwd_fuse_gpu import _chain, LEVELS_BIG
ops.set_precision("fp16")
N, groups = 3, 2
c = _chain(LEVELS_BIG, N, groups, seed=3 + N)
P, K, G = c["P"], c["K"], c["G"]
S = len(c["seg_rows"])
dxc = hip.conv2d_ml_fwd_bf16(c["gout"], c["wt"], LEVELS_BIG, N, k=3, pad=1, groups=groups)
dga_ref, dbe_ref = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
dx_ref = hip.groupnorm_relu_seg_bwd(dxc, c["y"], c["xa"], c["seg_rows"], c["mean"], c["rstd"], c["gamma"], dga_ref, dbe_ref, G, True, beta=c["beta"])
sr = hip._iarr(c["seg_rows"])
chunks = hip.load().utv2_groupnorm_seg_chunks(S, ctypes.cast(sr, hip.c_p))
ws = hip.workspace(1, c["xa"].device, "gn")
AB_ref = ws[chunks * K * 2: chunks * K * 2 + S * K * 2].clone().view(S, K, 2)
s12_ref = ws[chunks * K * 2 + S * K * 2: chunks * K * 2 + S * K * 2 + S * G * 2].clone().view(S, G, 2)
part = hip.gnb_part_buffer(P, K, "cuda")
gm = hip.conv2d_ml_fwd_bf16(c["gout"], c["wt"], LEVELS_BIG, N, k=3, pad=1, groups=groups, gnb=(c["bits"], c["xa"], part))
dga, dbe = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
dx = hip.groupnorm_seg_bwd_p64(gm, c["xa"], c["seg_rows"], c["mean"], c["rstd"], c["gamma"], dga, dbe, G, part)
AB = ws[: S * K * 2].clone().view(S, K, 2)
s12 = ws[S * K * 2: S * K * 2 + S * G * 2].clone().view(S, G, 2)
torch.cuda.synchronize()
r0 = 0
for s, n in enumerate(c["seg_rows"]):
    d = (dx[r0:r0 + n].float() - dx_ref[r0:r0 + n].float()).abs()
    print("seg", s, "rows", r0, r0 + n, "dx maxdiff %.3e of %.3e" % (float(d.max()), float(dx_ref[r0:r0 + n].float().abs().max())),
          "AB dev %.2e / %.2e" % (float((AB[s] - AB_ref[s]).abs().max()), float(AB_ref[s].abs().max())),
          "s12 dev %.2e / %.2e" % (float((s12[s] - s12_ref[s]).abs().max()), float(s12_ref[s].abs().max())))
    gd = gm[r0:r0 + n].double(); xh = (c["xa"][r0:r0 + n].double() - c["mean"][s].double().repeat_interleave(8)) * c["rstd"][s].double().repeat_interleave(8)
    Aex, Bex = (gd * xh).sum(0), gd.sum(0)
    print("     vs float64: fused A %.2e B %.2e | unfused A %.2e B %.2e" % (float((AB[s, :, 0].double() - Aex).abs().max()), float((AB[s, :, 1].double() - Bex).abs().max()),
          float((AB_ref[s, :, 0].double() - Aex).abs().max()), float((AB_ref[s, :, 1].double() - Bex).abs().max())))
    r0 += n
