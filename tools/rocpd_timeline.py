"""Timeline view of a rocprofv3 rocpd (.db) kernel trace: over the steady-state window (the last FRAC of the trace by time) the union
busy time of the GPU, the idle gaps, the per-queue busy time and the time by kernel class.
usage: rocpd_timeline.py DB [FRAC=0.5] [STEPS_IN_WINDOW]
       rocpd_timeline.py DB steps TOTAL_STEPS K     window = the last K of TOTAL_STEPS identical steps (ends of the sgd_momentum_f32 launches)"""
import re
import sqlite3
import sys

CLASSES = [("conv fwd/dgrad 256-tile", r"conv_igemm_bf16_(pp|rs|w8)"), ("conv fwd/dgrad 128-tile", r"conv_igemm_bf16"), ("conv wgrad 256-tile", r"conv_wgrad_bf16_(w8|pp)"),
           ("conv wgrad 128-tile", r"conv_wgrad_bf16"), ("wgrad reduce/colsum", r"reduce_slabs|colsum"), ("group norm", r"\d+gn_(stats|apply|bwd)"),
           ("ATen", r"at::native|at6native|rocclr"), ("losses/targets", r"focal|fcos_|iou|giou|loc_|smooth|ce_|softmax"),
           ("nms/topk/decode", r"nms|topk|decode|rank"), ("optimizer/ema/cast", r"sgd|ema|f32_to_bf16|flip_transpose"),
           ("roi/rpn", r"roi_|rpn|match_|anchor|sample")]


def main(path, frac=0.5, steps=None, total_steps=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select d.start, d.end, s.kernel_name, {('d.' + qcol) if qcol else '0'} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    w0 = t1 - (t1 - t0) * frac
    if total_steps:
        marks = [r[1] for r in rows if "sgd_momentum" in r[2]]
        cps = len(marks) // total_steps
        w0, t1 = marks[-steps * cps - 1], marks[-1]
        rows = [r for r in rows if r[1] <= t1]
    rows = [r for r in rows if r[0] >= w0]
    span = (t1 - w0) / 1e6
    busy = 0
    cur_end = rows[0][0]
    gaps = []
    for s, e, _, _ in rows:
        if s > cur_end:
            gaps.append(s - cur_end)
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    per = " (%.3f ms / step)" % (span / steps) if steps else ""
    print("window %.3f ms%s, %d dispatches, union busy %.3f ms (%.1f%%), idle %.3f ms in %d gaps" %
          (span, per, len(rows), busy / 1e6, 100 * busy / 1e6 / span, sum(gaps) / 1e6, len(gaps)))
    for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
        g = [x for x in gaps if lo <= x < hi]
        print("  gaps %6.0f-%-8.0f us: %6d  total %8.3f ms" % (lo / 1e3, hi / 1e3, len(g), sum(g) / 1e6))
    qs = {}
    for s, e, _, q in rows:
        qs[q] = qs.get(q, 0) + e - s
    for q, v in sorted(qs.items(), key=lambda kv: -kv[1]):
        print("  queue %s: kernel time %.3f ms (%.1f%% of window)" % (q, v / 1e6, 100 * v / 1e6 / span))
    tot = sum(e - s for s, e, _, _ in rows)
    by = {}
    for s, e, name, _ in rows:
        for cname, pat in CLASSES:
            if re.search(pat, name):
                break
        else:
            cname = "other"
        a = by.setdefault(cname, [0, 0])
        a[0] += 1; a[1] += e - s
    print("kernel time by class (sum %.3f ms = %.2fx the window: >1 means concurrency):" % (tot / 1e6, tot / 1e6 / span))
    for cname, (n, v) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("  %-26s %7d dispatches %9.3f ms %5.1f%%%s" % (cname, n, v / 1e6, 100.0 * v / tot, ("  %.3f ms/step" % (v / 1e6 / steps)) if steps else ""))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "steps":
        main(sys.argv[1], steps=int(sys.argv[4]), total_steps=int(sys.argv[3]))
    else:
        main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, int(sys.argv[3]) if len(sys.argv) > 3 else None)
