"""Per-kernel sums of a PMC counter from a rocprofv3 rocpd (.db) run (one --pmc pass)."""
import sqlite3
import sys


def pmc(path, top=12):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    names = dict(cur.execute(f"select id, name from {ip}").fetchall())
    rows = cur.execute(
        f"select s.kernel_name, e.pmc_id, count(*), sum(e.value), sum(d.end-d.start) from {pe} e "
        f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, e.pmc_id order by 4 desc").fetchall()
    out = ["# %s" % path, "%-70s %-12s %7s %16s %14s %12s" % ("kernel", "counter", "calls", "sum", "per_call", "kernel_ms")]
    for name, pid, n, v, t in rows[:top]:
        out.append("%-70s %-12s %7d %16.1f %14.2f %12.3f" % (name[:70], names.get(pid, pid), n, v, v / n, t / 1e6))
    return "\n".join(out)


if __name__ == "__main__":
    print(pmc(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12))
