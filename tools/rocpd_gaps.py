"""The largest all-queues-idle gaps of a rocprofv3 kernel trace (last K of TOTAL identical bench steps): what ran before and after each.
usage: rocpd_gaps.py DB TOTAL_STEPS K [TOP=40] [QUEUE]      QUEUE: only the dispatches of that queue id (its own idle time)"""
import sqlite3
import sys


def main(path, total_steps, k, top=40, queue=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    rows = cur.execute(f"select d.start, d.end, s.kernel_name, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    marks = [r[1] for r in rows if "sgd_momentum" in r[2]]
    cps = len(marks) // total_steps
    w0, w1 = marks[-k * cps - 1], marks[-1]
    rows = [r for r in rows if r[0] >= w0 and r[1] <= w1]
    if queue is not None:
        qs = sorted(set(r[3] for r in rows))
        print("queues:", qs)
        rows = [r for r in rows if str(r[3]) == str(queue)]
    gaps = []
    cur_end, last = rows[0][1], rows[0]
    for r in rows[1:]:
        if r[0] > cur_end:
            gaps.append((r[0] - cur_end, last, r))
        if r[1] > cur_end:
            cur_end, last = r[1], r
    agg = {}
    for g, a, b in gaps:
        key = (a[2][:48], b[2][:48])
        e = agg.setdefault(key, [0, 0])
        e[0] += 1; e[1] += g
    print("idle gaps by (kernel before -> kernel after), %d steps: total %.3f ms/step" % (k, sum(g for g, _, _ in gaps) / 1e6 / k))
    for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%7.1f us/step %5.1f x/step avg %6.1f us   %-48s -> %s" % (t / 1e3 / k, n / k, t / 1e3 / n, a, b))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 40, sys.argv[5] if len(sys.argv) > 5 else None)
