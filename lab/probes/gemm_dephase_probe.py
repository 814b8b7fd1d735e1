"""Does running a CU's two 4-wave workgroups in ANTI-PHASE (one in its k-loop while the other is in its epilogue) pay?
gemm_x3_kernel's 4-wave form (224 x 128 tiles, two persistent workgroups per CU, probe library) with a start delay for every CU's
second workgroup (mdm_debug_set(8, cycles / 64)); kernel-only timing at the four encoder shapes, plain and production epilogues.
Usage: python tools/gemm_dephase_probe.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdm_amd  # noqa: F401
from mdm_amd import _native

lib = _native.load_probe()
dev = "cuda:0"
NSEQ, S = 256, 197
M = NSEQ * S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
stream = torch.cuda.current_stream().cuda_stream


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [("in_proj", M, 1536, 512, 0, False), ("out_proj", M, 512, 512, 0, True), ("linear1", M, 1024, 512, 1, False),
          ("linear2", M, 512, 1024, 0, True)]
DELAYS_US = [0, 3, 6, 10, 15, 21, 30]
for name, m, n, k, act, res in shapes:
    a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    out = torch.empty(m, n, device=dev)
    nb = lib.mdm_linear_x3_scratch_bytes(m, n, k)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    lib.check(lib.mdm_linear_x3(a.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), m, n, k, 0, scratch.data_ptr(), nb,
                                stream), "x3")
    lib.mdm_debug_set(1, 1)
    variants = [("w8", 0)] + [("w4", d) for d in DELAYS_US]
    times = {v: [] for v in variants}
    for _ in range(5):
        for v in variants:
            lib.mdm_debug_set(2, int(v[0][1]))
            lib.mdm_debug_set(8, int(v[1] * 2000 / 64) if v[1] else 0)       # ~2.0 GHz under this load

            def run():
                lib.check(lib.mdm_linear_x3(a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                                            m, n, k, act, scratch.data_ptr(), nb, stream), "x3")
            times[v].append(timeit(run, reps))
    lib.mdm_debug_set(8, 0); lib.mdm_debug_set(2, 8); lib.mdm_debug_set(1, 0)
    for v in variants:
        ts = sorted(times[v])
        print(f"{name:9s} N={n} K={k} {v[0]} second-workgroup delay {v[1]:2d} us: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us", flush=True)
