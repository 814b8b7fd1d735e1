"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel --stats table we commit under profiles/.
Usage: python tools/rocpd_summary.py <results.db> [> profiles/<name>.md]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, cnt, tot, avg, mn, mx in rows:
        print(f"| `{n[:150]}` | {cnt} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
