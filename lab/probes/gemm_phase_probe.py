"""Where a tile's time goes inside the split-precision GEMM (ablation build ABL = 128: s_memtime counters of wave 0 of every
workgroup; include/mdm_hip.h mdm_debug_get).  Usage: python tools/gemm_phase_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdm_amd  # noqa: F401
from mdm_amd import _native

lib = _native.load_probe()     # the -DMDM_PROBES build (include/mdm_hip_probe.h)
dev = "cuda:0"
NSEQ, S = 256, 197
M = NSEQ * S
stream = torch.cuda.current_stream().cuda_stream
for name, n, k in (("in_proj", 1536, 512), ("out_proj", 512, 512), ("linear1", 1024, 512), ("linear2", 512, 1024)):
    a = torch.randn(M, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    out = torch.empty(M, n, device=dev)
    nb = lib.mdm_linear_x3_scratch_bytes(M, n, k)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)

    def run():
        lib.check(lib.mdm_linear_x3(a.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), M, n, k, 0,
                                        scratch.data_ptr(), nb, stream), "x3")
    lib.mdm_debug_set(2, 8)
    run()
    lib.mdm_debug_set(1, 1)
    lib.mdm_debug_set(0, 128)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.mdm_debug_get(-1, None)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    v = []
    for i in range(7):
        d = C.c_double()
        lib.mdm_debug_get(i, C.byref(d))
        v.append(d.value)
    lib.mdm_debug_set(0, 0)
    lib.mdm_debug_set(1, 0)
    tiles, steps = v[4], v[5]
    us = e0.elapsed_time(e1) / reps * 1e3
    # the counter's tick is calibrated on the launch time: a CU's tiles run back to back, so (k-loop + epilogue) ticks per tile
    # x tiles per CU = the launch
    per_cu = max(1.0, tiles / reps / 256.0)
    tick = us / per_cu / ((v[2] + v[3]) / tiles)
    print(f"{name:9s} {us:7.1f} us/launch, {per_cu:.1f} tiles/CU | per tile: k-loop {v[2] / tiles * tick:6.2f} us, epilogue "
          f"{v[3] / tiles * tick:6.2f} us | per k step: {v[2] / steps * tick:5.3f} us, of which vmcnt(0) wait "
          f"{(v[0] + v[1]) / steps * tick:5.3f} (first step of a tile {v[0] / tiles * tick:5.3f}) and barrier wait "
          f"{v[6] / steps * tick:5.3f}", flush=True)
