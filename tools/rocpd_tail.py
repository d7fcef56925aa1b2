"""The end of a training step in a rocprofv3 kernel trace: every dispatch of the last WINDOW_US microseconds before the optimizer's first
kernel (amp_found_inf / sgd_momentum), per queue - is the step's tail one stream running alone?
usage: rocpd_tail.py DB TOTAL_STEPS [STEP_FROM_END=2] [WINDOW_US=2500] [MARKER [AFTER_US]]
MARKER: a kernel-name substring to centre the window on instead of the optimizer (e.g. loss_combine: the forward / backward seam)"""
import sqlite3
import sys


def main(path, total_steps, from_end=2, window_us=2500.0, marker=None, after_us=0.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    rows = cur.execute(f"select d.start, d.end, s.kernel_name, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    if marker:
        opt = [r for r in rows if marker in r[2]]
    else:
        opt = [r for r in rows if "amp_found_inf" in r[2]] or [r for r in rows if "sgd_momentum" in r[2]]
    tm = opt[-from_end][0]
    t1 = tm + after_us * 1e3
    t0 = tm - window_us * 1e3
    sel = [r for r in rows if r[1] > t0 and r[0] < t1]
    qs = sorted(set(r[3] for r in sel))
    print("window %.0f us before the optimizer of step -%d; queues %s" % (window_us, from_end, qs))
    for q in qs:
        mine = [r for r in sel if r[3] == q]
        busy = sum(min(r[1], t1) - max(r[0], t0) for r in mine) / 1e3
        print("queue %s: %d dispatches, busy %.0f us, last end at -%.0f us" % (q, len(mine), busy, (t1 - max(r[1] for r in mine)) / 1e3))
    for r in sel:
        print("%9.1f %9.1f  q%-3s %7.1f us  %s" % ((r[0] - t1) / 1e3, (r[1] - t1) / 1e3, r[3], (r[1] - r[0]) / 1e3, r[2][:70]))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]), int(a[3]) if len(a) > 3 else 2, float(a[4]) if len(a) > 4 else 2500.0, a[5] if len(a) > 5 else None,
         float(a[6]) if len(a) > 6 else 0.0)
