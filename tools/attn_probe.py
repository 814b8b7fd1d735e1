"""Attention kernel probe at the headline shape (256 sequences x 197 tokens, 4 heads of 128), through the C ABI, timed
with events on the launch stream; kernel only (mdm_debug_set(1, 1): the operand planes of the first call are reused).
Usage: python tools/attn_probe.py [reps] [ablate,ablate,...]     codes: attention_x3.h ABL (mdm_debug_set(3, code))"""
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
codes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
stream = torch.cuda.current_stream().cuda_stream
qkv = torch.randn(M, 3 * D, device=dev)
qkv[:, :D] *= 128 ** -0.5
out = torch.empty(M, D, device=dev)
nb = lib.mdm_attention_x3_scratch_bytes(NSEQ, S, D)
scratch = torch.empty(nb, dtype=torch.uint8, device=dev)


def run():
    lib.check(lib.mdm_attention_x3(qkv.data_ptr(), out.data_ptr(), None, NSEQ, NSEQ, S, D, H, scratch.data_ptr(), nb,
                                       stream), "att3")


def timeit(n):
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


run()
lib.mdm_debug_set(1, 1)
times = {c: [] for c in codes}
for _ in range(5):
    for c in codes:
        lib.mdm_debug_set(3, c)
        times[c].append(timeit(reps))
lib.mdm_debug_set(3, 0)
lib.mdm_debug_set(1, 0)
fl = 4.0 * NSEQ * H * S * S * 128
for c in codes:
    ts = sorted(times[c])
    print(f"attention f16x3 ablate={c:3d}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us   {fl / ts[len(ts) // 2] / 1e6:6.1f} TF alg",
          flush=True)
