#!/bin/bash
# kernel trace of one bench loop at a given batch: bash tools/gpu_r4_trace.sh <tag> <B> [ENV=V ...]
set -u
TAG=${1:-r4trace}; B=${2:-1}; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --batch $B --steps 1 --warmup 1 --quick > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_summary.py $DB --by-grid > $OUT/kernel_stats.md
  python - $DB <<'PY' > $OUT/gaps.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select start, end from kernels order by start").fetchall()
rows = rows[len(rows) // 2:]           # the timed loop (second half: warm-up + profile pass come first / last)
busy = sum(e - s for s, e in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
gaps.sort()
print(f"launches {len(rows)} span {span/1e6:.3f} ms busy {busy/1e6:.3f} ms gaps total {sum(gaps)/1e6:.3f} ms median gap {gaps[len(gaps)//2]/1e3:.2f} us p90 {gaps[int(len(gaps)*0.9)]/1e3:.2f} us")
PY
  cat $OUT/gaps.txt; head -20 $OUT/kernel_stats.md | cut -c1-210; rm -f $DB
fi
find $OUT/prof -name '*.csv' -size +2M -delete
