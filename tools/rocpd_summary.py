"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel --stats table we commit under profiles/.
Usage: python tools/rocpd_summary.py <results.db> [--by-grid]"""
import sqlite3
import sys


def main(path, by_grid=False):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    grid = next((g for g in ("grid_x", "grid_size_x", "grid_size") if g in cols), None)
    key = f"{name}, {grid}" if (by_grid and grid) else name
    rows = c.execute(f"select {key}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {key} order by sum(end-start) desc").fetchall()
    total = sum(r[-4] for r in rows) or 1
    print("| kernel | " + ("grid | " if by_grid and grid else "") + "calls | total ms | avg us | min us | max us | % |")
    print("|---|" + ("---|" if by_grid and grid else "") + "---|---|---|---|---|---|")
    for row in rows:
        n = row[0]
        g = f"{row[1]} | " if (by_grid and grid) else ""
        cnt, tot, avg, mn, mx = row[-5:]
        print(f"| `{n[:110]}` | {g}{cnt} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], "--by-grid" in sys.argv)
