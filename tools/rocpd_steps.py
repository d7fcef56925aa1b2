"""Per-step view of a rocprofv3 rocpd (.db) kernel trace of a training loop: the window is the last K steps, a step ending with its last
optimizer launch (a run of sgd_momentum* dispatches followed by anything else), so no step count has to be passed in.
Prints ms / step, dispatches / step, union busy time, the idle-gap histogram and the per-kernel table (calls / step, average us, ms / step).
usage: rocpd_steps.py DB [K=20] [TOP=70]"""
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    return cur.execute(f"select d.start, d.end, s.kernel_name, {('d.' + qcol) if qcol else '0'} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()


def main(path, k=20, top=70):
    rows = load(path)
    ends = []       # end time of the last sgd launch of every step
    for i, r in enumerate(rows):
        if "sgd_momentum" in r[2] and (i + 1 == len(rows) or "sgd_momentum" not in rows[i + 1][2]):
            ends.append(r[1])
    if len(ends) < k + 1:
        k = len(ends) - 1
    w0, w1 = ends[-k - 1], ends[-1]
    rows = [r for r in rows if r[0] >= w0 and r[1] <= w1]
    span = (w1 - w0) / 1e6
    busy, cur_end, gaps = 0, rows[0][0], []
    for s, e, _, _ in rows:
        if s > cur_end:
            gaps.append(s - cur_end)
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    tot = sum(e - s for s, e, _, _ in rows)
    print("window: last %d steps, %.3f ms / step, %.1f dispatches / step, union busy %.3f ms / step (%.1f %%), kernel time %.3f ms / step (%.2fx the window)"
          % (k, span / k, len(rows) / k, busy / 1e6 / k, 100 * busy / 1e6 / span, tot / 1e6 / k, tot / 1e6 / span))
    print("all-queues-idle time %.3f ms / step in %.1f gaps / step" % (sum(gaps) / 1e6 / k, len(gaps) / k))
    for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 1e4), (1e4, 2e4), (2e4, 1e5), (1e5, 1e12)):
        g = [x for x in gaps if lo <= x < hi]
        print("  gaps %6.0f-%-8.0f us: %7.1f / step  %8.3f ms / step" % (lo / 1e3, hi / 1e3, len(g) / k, sum(g) / 1e6 / k))
    qs = {}
    for s, e, _, q in rows:
        a = qs.setdefault(q, [0, 0]); a[0] += 1; a[1] += e - s
    for q, (n, v) in sorted(qs.items(), key=lambda kv: -kv[1][1]):
        print("  queue %s: %.1f dispatches / step, kernel time %.3f ms / step" % (q, n / k, v / 1e6 / k))
    by = {}
    for s, e, name, _ in rows:
        a = by.setdefault(name, [0, 0]); a[0] += 1; a[1] += e - s
    print("%-100s %9s %9s %9s %6s" % ("kernel", "calls/st", "avg_us", "ms/step", "pct"))
    for name, (n, v) in sorted(by.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-100s %9.1f %9.2f %9.3f %5.1f%%" % (name[:100], n / k, v / n / 1e3, v / 1e6 / k, 100.0 * v / tot))
    short = [(e - s) for s, e, _, _ in rows if e - s < 10e3]
    print("dispatches shorter than 10 us: %.1f / step, %.3f ms / step" % (len(short) / k, sum(short) / 1e6 / k))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 70)
