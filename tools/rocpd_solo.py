"""Which kernels run ALONE (no other queue active) and for how long?  Solo time of a kernel that cannot fill the chip is the waste a second
stream could take.  Last K of TOTAL identical bench steps of a rocprofv3 kernel trace.   usage: rocpd_solo.py DB TOTAL_STEPS K [TOP=30]"""
import sqlite3
import sys


def main(path, total_steps, k, top=30):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    grid = "grid_size_x" if "grid_size_x" in cols else None
    wg = "workgroup_size_x" if "workgroup_size_x" in cols else None
    extra = (", d.%s, d.%s" % (grid, wg)) if grid and wg else ", 0, 0"
    rows = cur.execute(f"select d.start, d.end, s.kernel_name, d.{qcol}{extra} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    marks = [r[1] for r in rows if "sgd_momentum" in r[2]]
    cps = len(marks) // total_steps
    w0, w1 = marks[-k * cps - 1], marks[-1]
    rows = [r for r in rows if r[0] >= w0 and r[1] <= w1]
    ev = []
    for i, r in enumerate(rows):
        ev.append((r[0], 1, i)); ev.append((r[1], -1, i))
    ev.sort()
    active = set()
    solo = {}
    last = None
    for t, d, i in ev:
        if last is not None and len(active) == 1 and t > last:
            j = next(iter(active))
            e = solo.setdefault(rows[j][2], [0.0, 0, 0.0, 0])
            e[0] += t - last
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
        last = t
    tot = {}
    for r in rows:
        e = tot.setdefault(r[2], [0.0, 0, 0])
        e[0] += r[1] - r[0]; e[1] += 1
        e[2] = max(e[2], (r[4] // r[5]) if r[5] else 0)
    print("solo time (exactly one kernel in flight) per step, %d steps; window %.2f ms/step" % (k, (w1 - w0) / 1e6 / k))
    print("%10s %10s %7s %9s  kernel" % ("solo us", "total us", "calls", "max WGs"))
    for name, e in sorted(solo.items(), key=lambda kv: -kv[1][0])[:top]:
        t = tot[name]
        print("%10.1f %10.1f %7.1f %9d  %s" % (e[0] / 1e3 / k, t[0] / 1e3 / k, t[1] / k, t[2], name[:80]))
    print("total solo %.3f ms/step" % (sum(e[0] for e in solo.values()) / 1e6 / k))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]), int(a[3]), int(a[4]) if len(a) > 4 else 30)
