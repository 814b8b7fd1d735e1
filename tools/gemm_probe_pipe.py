"""Decomposition of the encoder GEMM's time at the headline shapes (256 sequences x 197 tokens), plain fp32-out epilogue, through the
probe library: the step-synchronous k-loop vs the pipelined one (mdm_debug_set(6, 1)), each with the ablation codes
0 production, 1 no epilogue stores, 2 no loads after the prologue, 4 no MFMAs and their sums (3, 5, 6, 7).
Kernel-only timing (planes reused), interleaved rounds.  Usage: python tools/gemm_probe_pipe.py [reps] [codes]"""
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
codes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 5, 6, 7]
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


lib.mdm_debug_set(2, 8)
for name, n, k in [("in_proj", 1536, 512), ("out_proj", 512, 512), ("linear1", 1024, 512), ("linear2", 512, 1024)]:
    a = torch.randn(M, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    if os.environ.get("PROBE_ZERO"):
        a.zero_(); w.zero_()
    b = torch.randn(n, device=dev)
    out = torch.empty(M, n, device=dev)
    nb = lib.mdm_linear_x3_scratch_bytes(M, n, k)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)

    def run():
        lib.check(lib.mdm_linear_x3(a.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), M, n, k, 0,
                                    scratch.data_ptr(), nb, stream), "x3")
    lib.mdm_debug_set(6, 2); lib.mdm_debug_set(0, 0); lib.mdm_debug_set(1, 0)
    run()
    ref = out.clone()
    lib.mdm_debug_set(6, 1)
    run()
    torch.cuda.synchronize()
    print(f"{name}: pipelined vs step-synchronous result max-abs diff {float((out - ref).abs().max()):.3e} "
          f"(|out| max {float(ref.abs().max()):.2f})", flush=True)
    lib.mdm_debug_set(1, 1)
    variants = [(p, c) for p in (2, 1) for c in codes]
    times = {v: [] for v in variants}
    for _ in range(5):
        for v in variants:
            lib.mdm_debug_set(6, v[0]); lib.mdm_debug_set(0, v[1])
            times[v].append(timeit(run, reps))
    lib.mdm_debug_set(6, 0); lib.mdm_debug_set(0, 0); lib.mdm_debug_set(1, 0)
    for v in variants:
        ts = sorted(times[v])
        print(f"{name:9s} N={n} K={k} {'PIPE' if v[0] == 1 else 'sync'} ablate={v[1]}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us  "
              f"{2 * M * n * k / ts[len(ts) // 2] / 1e6:6.1f} TF alg", flush=True)
