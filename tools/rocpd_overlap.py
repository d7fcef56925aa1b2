"""What runs on the OTHER queues while a given kernel runs?  For every dispatch of kernels matching PATTERN (last K of TOTAL identical bench
steps of a rocprofv3 kernel trace): its duration and the kernels overlapping it, aggregated by name.
usage: rocpd_overlap.py DB TOTAL_STEPS K PATTERN [SHOW=6]   (SHOW: also print the first SHOW dispatches one by one)"""
import re
import sqlite3
import sys


def main(path, total_steps, k, pattern, show=6):
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
    pat = re.compile(pattern)
    agg, n, tot = {}, 0, 0.0
    for t in [r for r in rows if pat.search(r[2])]:
        n += 1
        dur = (t[1] - t[0]) / 1e3
        tot += dur
        line = []
        for o in rows:
            if o[3] == t[3] or o[1] <= t[0] or o[0] >= t[1]:
                continue
            ov = (min(o[1], t[1]) - max(o[0], t[0])) / 1e3
            nm = re.sub(r"^_Z\d+", "", o[2])[:44]
            e = agg.setdefault(nm, [0, 0.0])
            e[0] += 1; e[1] += ov
            line.append("%s %.0f/%.0f us" % (nm, ov, (o[1] - o[0]) / 1e3))
        if n <= show:
            print("%7.1f us  q%s  | %s" % (dur, t[3], "; ".join(line) or "alone"))
    print("%d dispatches of /%s/, avg %.1f us; overlapped by:" % (n, pattern, tot / max(n, 1)))
    for nm, (c, ov) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-44s %5.2f x/dispatch  %7.1f us of overlap per dispatch" % (nm, c / n, ov / n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 6)
