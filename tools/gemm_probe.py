"""Kernel probes at the headline shapes (256 sequences x 197 tokens), through the C ABI, timed with events on the launch
stream.  Usage: python tools/gemm_probe.py [reps] [ablate,ablate,...]
  * the four encoder GEMM shapes on the f16x3 kernel, per ablation code (mdm_debug_set(0, code): 0 production,
    1 no epilogue stores, 2 no loads after the prologue, 4 no MFMAs, 8 LDS-DMA issued as a burst; codes other than 0
    only exist for the plain fp32-out variant, so every shape is run as (act none, no residual) under ablation);
  * attention: exact-fp32 kernel vs the split-precision kernel (the latter timed without its test-only pack kernel
    by timing pack alone and subtracting is NOT done -- the x3 number includes qkv_pack; see the model-level
    kernel_ms in bench.py for the in-situ figure).
The GEMM timings are kernel-only: after one normal call, mdm_debug_set(1, 1) makes mdm_linear_x3 reuse the operand
planes already in its scratch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdm_amd  # noqa: F401
from mdm_amd import _native

lib = _native.load_probe()     # the -DMDM_PROBES build (include/mdm_hip_probe.h)
dev = "cuda:0"
NSEQ, S, D, H = 256, 197, 512, 4
M = NSEQ * S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ablates = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
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
    return a.elapsed_time(b) / n * 1e3  # us


ROUNDS = 5
shapes = [("in_proj", M, 1536, 512, 0, False), ("out_proj", M, 512, 512, 0, True), ("linear1", M, 1024, 512, 1, False),
          ("linear2", M, 512, 1024, 0, True)]
for name, m, n, k, act, res in shapes:
    a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    if os.environ.get("PROBE_ZERO"):   # DVFS check: zero operands draw less power -> higher clock at identical work
        a.zero_(); w.zero_()
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    out = torch.empty(m, n, device=dev)
    nb = lib.mdm_linear_x3_scratch_bytes(m, n, k)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    lib.check(lib.mdm_linear_x3(a.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), m, n, k, 0,
                                    scratch.data_ptr(), nb, stream), "x3")   # fills the planes
    lib.mdm_debug_set(1, 1)
    # variants: every ablation code on the plain epilogue (act none, no residual) + the production epilogue of this shape
    waves_list = [int(x) for x in os.environ.get("PROBE_WAVES", "4").split(",")]
    variants = [(f"w{wv}", ab, 0, False) for wv in waves_list for ab in ablates] + \
               ([(f"w{wv}p", 0, act, res) for wv in waves_list] if (act or res) else [])
    times = {v: [] for v in variants}
    for _ in range(ROUNDS):          # interleaved rounds: the chip's clock drifts with its power state (DVFS), so
        for v in variants:           # back-to-back blocks per variant would measure the drift, not the kernel
            tag, ab, use_act, use_res = v
            lib.mdm_debug_set(0, ab)
            lib.mdm_debug_set(2, int(tag[1]))

            def run():
                lib.check(lib.mdm_linear_x3(a.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                r.data_ptr() if use_res else None, out.data_ptr(), m, n, k, use_act,
                                                scratch.data_ptr(), nb, stream), "x3")
            times[v].append(timeit(run, reps))
    lib.mdm_debug_set(0, 0)
    lib.mdm_debug_set(1, 0)
    lib.mdm_debug_set(2, 4)
    # the same shape on the f16f6 k-loop (gemm_f16f6.h on this kernel's skeleton, 8 waves), production epilogue of the shape
    # where it exists (act none +- res, gelu without res), kernel-only like the rows above
    nb6 = lib.mdm_linear_f16f6_scratch_bytes(m, n, k)
    scratch6 = torch.empty(nb6, dtype=torch.uint8, device=dev)
    lib.mdm_debug_set(2, 8)

    def run6(use_act, use_res):
        lib.check(lib.mdm_linear_f16f6(a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if use_res else None,
                                       out.data_ptr(), m, n, k, use_act, scratch6.data_ptr(), nb6, stream), "f16f6")
    run6(0, False)
    lib.mdm_debug_set(1, 1)
    f6v = [("f16f6", 0, False)] + ([("f16f6p", act, res)] if (act or res) else []) + [("f16f6w4", 0, False)]
    t6 = {v: [] for v in f6v}
    tb = []
    for _ in range(ROUNDS):
        for v in f6v:
            lib.mdm_debug_set(2, 4 if v[0].endswith("w4") else 8)   # f16f6w4: two independent 4-wave workgroups per CU
            t6[v].append(timeit(lambda: run6(v[1], v[2]), reps))
        lib.mdm_debug_set(2, 8)
        lib.mdm_debug_set(0, 0)
        tb.append(timeit(lambda: lib.check(lib.mdm_linear_x3(a.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), m, n, k, 0,
                                                                   scratch.data_ptr(), nb, stream), "x3"), reps))
    lib.mdm_debug_set(1, 0)
    lib.mdm_debug_set(2, 4)
    tb = sorted(tb)[len(tb) // 2]
    for v in f6v:
        ts = sorted(t6[v])
        med = ts[len(ts) // 2]
        print(f"{name:9s} N={n} K={k} {v[0]:6s} act={v[1]} res={int(v[2])}: median {med:7.1f} us  min {ts[0]:7.1f} us "
              f"{2 * m * n * k / med / 1e6:6.1f} TF alg   (f16x3 plain, interleaved: {tb:7.1f} us -> {tb / med:4.2f}x)", flush=True)
    for v in variants:
        ts = sorted(times[v])
        med, mn = ts[len(ts) // 2], ts[0]
        print(f"{name:9s} N={n} K={k} {v[0]:5s} ablate={v[1]:3d} act={v[2]} res={int(v[3])}: median {med:7.1f} us  min {mn:7.1f} us "
              f"{2 * m * n * k / med / 1e6:6.1f} TF alg", flush=True)

# attention
qkv = torch.randn(M, 3 * D, device=dev)
qkv[:, :D] *= 128 ** -0.5
out = torch.empty(M, D, device=dev)
us32 = timeit(lambda: lib.check(lib.mdm_attention(qkv.data_ptr(), out.data_ptr(), None, NSEQ, NSEQ, S, D, H, stream),
                                "att"), reps)
nb = lib.mdm_attention_x3_scratch_bytes(NSEQ, S, D)
scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
us3 = timeit(lambda: lib.check(lib.mdm_attention_x3(qkv.data_ptr(), out.data_ptr(), None, NSEQ, NSEQ, S, D, H,
                                                         scratch.data_ptr(), nb, stream), "att3"), reps)
fl = 4.0 * NSEQ * H * S * S * 128
print(f"attention f32   : {us32:8.1f} us  {fl / us32 / 1e6:6.1f} TF alg")
print(f"attention f16x3: {us3:8.1f} us  (includes the test-only qkv_pack kernel)")
