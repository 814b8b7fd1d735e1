"""Aggregate rocprofv3 --pmc results (rocpd sqlite) per kernel (+grid): mean counter value per dispatch.
Usage: python tools/rocpd_pmc.py <results.db> [name-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=""):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, grid_size, counter_name, value, dispatch_id from counters_collection").fetchall()
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for name, grid, cn, val, did in rows:
        if filt and filt not in name:
            continue
        per[(name.split("(")[0][-100:], grid)][cn][did] += val      # sum over instances of one dispatch
    for key, counters in sorted(per.items(), key=lambda kv: -len(kv[1])):
        nd = max(len(v) for v in counters.values())
        print(f"== {key[0]} grid={key[1]} dispatches={nd}")
        for cn, d in sorted(counters.items()):
            vals = list(d.values())
            print(f"   {cn:32s} mean {sum(vals) / len(vals):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
