"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE PMC summaries (tools/rocpd_pmc.py output) and the kernel-trace summary
(tools/rocpd_stats.py output) of one command.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: 16-byte-per-lane reads are
counted at half their bytes on gfx950; our own gn_apply_relu, which reads exactly what it writes, shows FETCH = WRITE / 2).  Per kernel:
HBM bytes per launch = 2 * FETCH + WRITE, GB/s = bytes / the kernel-trace average duration (the PMC passes themselves run slower).
Two aggregate keys are what bench.py reads: the FCOS tower conv launch (conv_igemm_bf16_rs<true> - before round 5 _pp<true> - on whole rounds of 256 x 256 tiles +
conv_igemm_bf16_v2<128,true,64> on the remaining rows: one of each per launch) and the tower weight gradient (conv_wgrad_bf16_w8).
usage: make_traffic.py FETCH.txt WRITE.txt KERNEL_STATS.txt OUT.json"""
import json
import sys

# DF16b = __bf16 (libutv2_hip.so), DF16_ = _Float16 (libutv2_hip_f16.so: the same kernels on the fp16 build)
# (round 5: the multi-level 3x3 launches run on the row-span form conv_igemm_bf16_rs; earlier traces / UTV2_PP_RS=0: conv_igemm_bf16_pp)
# (round 6: the row-span kernel has a third template argument - Lb0E: the plain instantiation, Lb1E: the GroupNorm-backward dgrad form)
TOWERS = ([("_Z18conv_igemm_bf16_rsILb1EDF16%sLb0EEv10ConvArgs16" % t, "") for t in ("_", "b")] +
          [("_Z18conv_igemm_bf16_%sILb1EDF16%sEv10ConvArgs16" % (k, t), "_Z18conv_igemm_bf16_v2ILi128ELb1ELi64EDF16%sEv10ConvArgs16" % t)
           for k in ("rs", "pp") for t in ("_", "b")])
TOWERS_GNB = ["_Z18conv_igemm_bf16_rsILb1EDF16%sLb1EEv10ConvArgs16" % t for t in ("_", "b")]
WGRADS = ["_Z18conv_wgrad_bf16_pp11Wgrad16Args", "_Z18conv_wgrad_bf16_w811Wgrad16Args"]   # round 5: the ping-pong schedule; before / UTV2_WGRAD_PP=0: lock-step


def per_kernel(path):
    out = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 6 and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            out[f[0].replace(".kd", "")] = (int(f[2]), float(f[3]))   # calls, sum in KiB
    return out


def durations(path):
    out = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 7 and f[0].startswith("_Z") and f[1].isdigit():
            out[f[0].replace(".kd", "")] = (int(f[1]), float(f[3]))   # calls, avg_us
    return out


fetch, write, dur = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), durations(sys.argv[3])
res = {"fetch_correction": 2.0, "source": "%s, %s, %s" % tuple(sys.argv[1:4])}
TOWER = next((t for t in TOWERS if t[0] in fetch), TOWERS[0])
if TOWER[0] in fetch:
    # the 256-tile kernel's share of the launch (whole rounds of 256 x 256 tiles: 97.5 % of the student's rows, 73 % of the teacher's); the
    # remaining rows run on conv_igemm_bf16_v2<128,true,64>, whose PMC row cannot be split from the head prediction convs that use
    # the same instance, so it is left out here (r01 added ALL of that kernel's bytes to this figure)
    n = fetch[TOWER[0]][0]
    fk = fetch[TOWER[0]][1] / n
    wk = write[TOWER[0]][1] / n
    res["tower_conv"] = {
        "kernel": TOWER[0],
        "fetch_size_kib_per_launch": round(fk, 2), "write_size_kib_per_launch": round(wk, 2),
        "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "launches_in_pmc_run": n,
        "covers": "the 256-tile kernel only (conv_igemm_bf16_rs<true> / _pp<true>: the whole rounds of 256x256 tiles of each launch)"}
for g in TOWERS_GNB:
    if g in fetch and g in write:
        n = fetch[g][0]
        res["tower_conv_gnb"] = {
            "kernel": g, "fetch_size_kib_per_launch": round(fetch[g][1] / n, 2), "write_size_kib_per_launch": round(write[g][1] / n, 2),
            "hbm_bytes_per_launch": (2.0 * fetch[g][1] / n + write[g][1] / n) * 1024.0, "launches_in_pmc_run": n,
            "covers": "the tower dgrads that also apply the ReLU mask plane and leave GroupNorm backward's partial sums (DESIGN 10.7): + the GroupNorm input read"}
kernels = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, (1, 0))[1] + write.get(k, (1, 0))[1])):
    nf, f = fetch.get(k, (0, 0.0))
    nw, w = write.get(k, (0, 0.0))
    n = max(nf, nw, 1)
    b = (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0
    if b < (1 << 20):
        continue
    e = {"launches_in_pmc_run": n, "fetch_size_kib_per_launch": round(f / max(nf, 1), 2), "write_size_kib_per_launch": round(w / max(nw, 1), 2),
         "hbm_bytes_per_launch": b}
    if k in dur:
        e["avg_us_kernel_trace"] = dur[k][1]
        e["hbm_GBps"] = round(b / dur[k][1] / 1e3, 1)
    kernels[k] = e
res["kernels"] = kernels
for WGRAD in WGRADS:
    if WGRAD in kernels:
        res["conv_wgrad_bf16_w8"] = dict(kernels[WGRAD], kernel=WGRAD)   # (key name kept: bench.py reads it)
        break
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
for k, e in list(kernels.items())[:25]:
    print("%-80s %8.1f MiB/launch %8s GB/s" % (k[:80], e["hbm_bytes_per_launch"] / 2 ** 20, e.get("hbm_GBps", "-")))
