"""Board power / shader clock while (a) the paired tower launch loops alone, (b) an HBM-bound pass loops alone, (c) the chip idles.
Reads the amdgpu hwmon files (power1_average / power1_input, power1_cap, freq1_input) at ~50 Hz from a thread; falls back to one
`rocm-smi --showpower --showmaxpower --showclocks --json` call per phase when hwmon is not there.  Evidence for the roofline line's
`sustained_clock_ghz`: is the clock held down by the power cap while MFMA kernels run?
usage: python tools/power_probe.py [seconds per phase]"""
import glob, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def hwmon_dirs():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if any(os.path.exists(os.path.join(d, f)) for f in ("power1_average", "power1_input")):
            out.append(d)
    return out


def rd(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    """samples EVERY card's hwmon (a one-GPU lease on an eight-GPU host still shows all eight in sysfs; the card under test is the
    one whose power moves with the load - picked per phase as the card with the highest mean power)"""

    def __init__(self, dirs):
        super().__init__(daemon=True)
        self.dirs, self.stop, self.rows = dirs, False, []
        self.files = []
        for d in dirs:
            pf = os.path.join(d, "power1_average")
            if not os.path.exists(pf):
                pf = os.path.join(d, "power1_input")
            self.files.append((pf, os.path.join(d, "freq1_input")))

    def run(self):
        while not self.stop:
            self.rows.append((time.time(), [(rd(pf), rd(ff)) for pf, ff in self.files]))
            time.sleep(0.02)


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=30).stdout
        return json.loads(o[o.index("{"):])
    except Exception as e:
        return {"error": repr(e)}


def phase(name, fn, dirs):
    fn(); torch.cuda.synchronize()
    s = Sampler(dirs) if dirs else None
    if s:
        s.start()
    t0 = time.time(); n = 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    mid = None
    while time.time() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
        if mid is None and not dirs and time.time() - t0 > SECS / 2:
            mid = smi()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / max(n, 1)
    rec = {"phase": name, "launches": n, "ms_per_launch": ms}
    if s:
        s.stop = True; s.join()
        rows = [r for r in s.rows if r[0] - t0 > SECS * 0.4]  # the settled part
        means = []
        for ci in range(len(dirs)):
            pw = [r[1][ci][0] for r in rows if r[1][ci][0] is not None]
            means.append(sum(pw) / len(pw) / 1e6 if pw else 0.0)
        ci = max(range(len(dirs)), key=lambda i: means[i])
        pw = [r[1][ci][0] for r in rows if r[1][ci][0] is not None]; fq = [r[1][ci][1] for r in rows if r[1][ci][1] is not None]
        rec["card"] = dirs[ci].split("/")[4]
        rec["all_cards_power_w_mean"] = [round(m, 1) for m in means]
        if pw:
            rec["power_w_mean"] = sum(pw) / len(pw) / 1e6; rec["power_w_max"] = max(pw) / 1e6
        if fq:
            rec["sclk_mhz_mean"] = sum(fq) / len(fq) / 1e6; rec["sclk_mhz_min"] = min(fq) / 1e6; rec["sclk_mhz_max"] = max(fq) / 1e6
        rec["samples"] = len(rows)
    elif mid is not None:
        rec["rocm_smi"] = mid
    return rec


def main():
    dirs = hwmon_dirs()
    info = {"hwmon": dirs}
    for d in dirs[:1]:
        for f in ("power1_cap", "power1_cap_max", "power1_cap_default"):
            v = rd(os.path.join(d, f))
            if v is not None:
                info[f + "_w"] = v / 1e6
    print(json.dumps(info))
    hip.set_h16(os.environ.get("UTV2_H16_KIND", "fp16"))
    H = hip.h16_dtype()
    N = 12
    level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    P = N * sum(h * w for h, w in level_hw)
    C, K = 256, 512
    x = torch.relu(torch.randn(P, C, device="cuda")).to(H)
    w16 = (torch.randn(K, 9 * C, device="cuda") * 0.05).to(H)
    y = torch.empty(P, K, device="cuda", dtype=H)
    fl = 2.0 * P * K * 9 * C
    a = torch.randn(64 << 20, device="cuda"); b = torch.randn(64 << 20, device="cuda")

    def tower():
        hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, k=3, pad=1, out=y)

    def ema():
        hip.ema_axpby(a, b, 0.9996)

    recs = [phase("idle", lambda: time.sleep(0.01), dirs), phase("tower 256->512 3x3, 12 images (MFMA-bound)", tower, dirs),
            phase("ema axpby 2 x 256 MB (HBM-bound)", ema, dirs), phase("tower again", tower, dirs)]
    for r in recs:
        if r["phase"].startswith("tower"):
            r["tflops"] = fl / r["ms_per_launch"] / 1e9
        if r["phase"].startswith("ema"):
            r["GBps"] = 3 * a.numel() * 4 / r["ms_per_launch"] / 1e6
        print(json.dumps(r))
    g = hip.conv_clock_probe() if hasattr(hip, "conv_clock_probe") else None
    print(json.dumps({"conv_clock_probe_ghz_us": g}))


if __name__ == "__main__":
    main()
