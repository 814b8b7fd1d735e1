"""Run the four encoder GEMM shapes of the headline config through the C ABI (for rocprofv3 kernel traces / PMC)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdm_amd
from mdm_amd import _native

lib = _native.load_native()
dev = "cuda:0"
M = 256 * 197
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else "x3"
ablate = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib.mdm_debug_set(0, ablate)
shapes = [(M, 1536, 512, 0, False), (M, 512, 512, 0, True), (M, 1024, 512, 1, False), (M, 512, 1024, 0, True)]
s = torch.cuda.current_stream().cuda_stream
for (m, n, k, act, res) in shapes:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev) if res else None
    out = torch.empty(m, n, device=dev)
    nb = lib.mdm_linear_bf16x3_scratch_bytes(m, n, k)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    def run():
        if which == "x3":
            lib.check(lib.mdm_linear_bf16x3(a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None,
                                            out.data_ptr(), m, n, k, act, scratch.data_ptr(), nb, s), "x3")
        else:
            lib.check(lib.mdm_linear(a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None,
                                     out.data_ptr(), m, n, k, act, s), "f32")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"ablate={ablate} {which} M={m} N={n} K={k} act={act} res={res}: {dt*1e6:.1f} us/call incl. operand split, {2*m*n*k/dt/1e12:.1f} TF algorithmic")
