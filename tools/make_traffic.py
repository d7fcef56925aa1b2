"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE PMC summaries (tools/rocpd_pmc.py output) for the dominant launch of
bench.py: the FCOS tower conv = conv_igemm_bf16_w8<true,__bf16> (whole rounds of 256x256 tiles) + conv_igemm_bf16_v2<128,true,64,__bf16>
(remaining rows), one of each per launch.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: 16-byte-per-lane reads are under-counted 2x on
gfx950).  usage: make_traffic.py FETCH.txt WRITE.txt OUT.json"""
import json
import sys

KERNELS = ("_Z18conv_igemm_bf16_w8ILb1EDF16bEv10ConvArgs16", "_Z18conv_igemm_bf16_v2ILi128ELb1ELi64EDF16bEv10ConvArgs16")


def per_kernel(path):
    out = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 6 and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            out[f[0].replace(".kd", "")] = (int(f[2]), float(f[3]))
    return out


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
n = fetch[KERNELS[0]][0]  # launches of the tower conv = dispatches of the w8 kernel
fk = sum(fetch[k][1] for k in KERNELS if k in fetch) / n
wk = sum(write[k][1] for k in KERNELS if k in write) / n
key = "conv_igemm_bf16_w8<true,__bf16>+conv_igemm_bf16_v2<128,true,64,__bf16>"
json.dump({key: {"fetch_size_kib_per_launch": round(fk, 2), "write_size_kib_per_launch": round(wk, 2), "fetch_correction": 2.0,
                 "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "launches_in_pmc_run": n,
                 "source": "%s, %s" % (sys.argv[1], sys.argv[2])}}, open(sys.argv[3], "w"), indent=1)
print(open(sys.argv[3]).read())
