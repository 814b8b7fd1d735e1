"""Inter-kernel gaps of a rocprofv3 (rocpd sqlite) kernel trace: for consecutive kernels on one queue / stream,
gap = start(n+1) - end(n).  Only pairs whose two kernels are both `mdm::` kernels and whose gap is below `--max-us` (default 50:
larger ones are host-side pauses between loops) are counted.  Prints the distribution and gap / (gap + kernel) shares.
Usage: python tools/rocpd_gaps.py <results.db> [--max-us 50]"""
import sqlite3
import sys


def main(path, max_us=50.0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, start, end from kernels order by start").fetchall()
    gaps, kern, pairs = [], 0.0, {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        if "mdm::" not in n0 or "mdm::" not in n1:
            continue
        g = (s1 - e0) / 1e3
        if g > max_us:
            continue
        gaps.append(g)
        kern += (e1 - s1) / 1e3
        k = (n0.split("(")[0][-48:], n1.split("(")[0][-48:])
        p = pairs.setdefault(k, [0, 0.0])
        p[0] += 1
        p[1] += g
    if not gaps:
        print("no mdm:: kernel pairs found")
        return
    gaps.sort()
    n = len(gaps)
    q = lambda f: gaps[min(n - 1, int(f * n))]      # noqa: E731
    tot = sum(gaps)
    print(f"pairs {n}: gap us min {gaps[0]:.2f} p10 {q(0.1):.2f} median {q(0.5):.2f} mean {tot / n:.2f} p90 {q(0.9):.2f} max {gaps[-1]:.2f} "
          f"(negative = overlap); sum of gaps {tot / 1e3:.3f} ms vs kernel time {kern / 1e3:.3f} ms = {100 * tot / (tot + kern):.1f} % of the chain")
    print("| predecessor -> successor | pairs | mean gap us |")
    print("|---|---|---|")
    for k, (cnt, s) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"| `{k[0]}` -> `{k[1]}` | {cnt} | {s / cnt:.2f} |")


if __name__ == "__main__":
    mx = float(sys.argv[sys.argv.index("--max-us") + 1]) if "--max-us" in sys.argv else 50.0
    main(sys.argv[1], mx)
