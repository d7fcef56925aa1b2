"""Summarise a rocprofv3 rocpd (.db) kernel trace into a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def stats(path, skip_first_frac=0.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    t0, t1 = cur.execute(f"select min(start), max(end) from {kd}").fetchone()
    out = ["# kernel-trace summary of %s" % path,
           "# total kernel time %.3f ms over %d dispatches; first-to-last span %.3f ms" % (total / 1e6, sum(r[1] for r in rows), (t1 - t0) / 1e6),
           "%-90s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for name, n, tot, mn, mx in rows:
        out.append("%-90s %8d %12.3f %10.2f %10.2f %10.2f %6.2f%%" % (name[:90], n, tot / 1e6, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    return "\n".join(out)


if __name__ == "__main__":
    print(stats(sys.argv[1]))
