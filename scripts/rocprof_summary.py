#!/usr/bin/env python
"""Turn a rocprofv3 (--kernel-trace --stats) results .db into the per-kernel text summary kept under profiles/."""
import sqlite3
import sys


def main(db, out=None, header=""):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                     "max(end-start)/1e3, max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [header, f"total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s} {'vgpr':>5s} {'lds':>7s}"]
    for r in rows:
        lines.append(f"{r[0][:78]:78s} {r[1]:7d} {r[2]:10.3f} {r[3]:9.1f} {r[4]:8.1f} {r[5]:9.1f} {100 * r[2] / tot:6.2f} {r[6]:5d} {r[7]:7d}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
