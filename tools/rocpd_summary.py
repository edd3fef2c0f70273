"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into the per-kernel stats table committed under profiles/."""
import sqlite3
import sys


def summarise(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["| kernel | calls | total ms | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name.split("(")[0][:70]
        out.append(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e6:.4f} | {mn / 1e6:.4f} | {mx / 1e6:.4f} | {100.0 * tot / total:.1f} |")
    return "\n".join(out), cols


if __name__ == "__main__":
    t, cols = summarise(sys.argv[1])
    print(t)
