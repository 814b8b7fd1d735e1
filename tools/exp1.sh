set -u
mkdir -p gpurun_out/exp1
for b in 32 64 128; do python bench.py --batch $b --no-cpu-baseline --steps 2 > gpurun_out/exp1/bench_b$b.json 2>&1; done
for a in 0 1 2 4 5 6; do python tools/gemm_probe.py 20 x3 $a > gpurun_out/exp1/probe_a$a.txt 2>&1; done
python -m pytest tests -m gpu -x -q -k progressive 2>&1 | tail -3
cat gpurun_out/exp1/probe_a*.txt
grep -h -o '"value": [0-9.]*\|"kernel_ms": {[^}]*}' gpurun_out/exp1/bench_b*.json
