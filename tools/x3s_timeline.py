"""Where does a gemm_x3s_kernel launch spend its time?  Shader-clock stamps of wave 0 of every workgroup of ONE selected launch
(probe library: csrc/gemm_x3s.h X3S_STAMP; mdm_debug_set(9, n) / mdm_debug_get(100 + ...)): kernel entry, first chunk visible,
k-loop retired, last store issued.  Runs one guided DiP forward at B motions (2 B sequences of 20 + 40 tokens) and, per selected
launch of layer 1, prints the distribution of the three phases over the workgroups.
Usage: python tools/x3s_timeline.py [B=32] [enc]        (enc: the encoder at B motions, T = 196, instead of DiP)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from helpers import make_pair, synth_dip_state_dict, synth_dip_y, synth_state_dict, synth_y, to_dev
from mdm_amd import _native

lib = _native.load_probe()
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
enc = len(sys.argv) > 2 and sys.argv[2] == "enc"
if enc:
    model, _ = make_pair(synth_state_dict(seed=0), 50, dev, guided=True, native_lib=lib)
    y = to_dev(synth_y(B, 196, seed=1, lengths=[196] * B), dev)
    x = torch.randn(B, 263, 1, 196, device=dev)
    # encoder layer 1: in_proj (kind 0), out_proj (2), linear1 (3), linear2 (2); layer 0 has 4 launches, InputProcess 1 in front
    names = {5: "in_proj (norm2 folded -> Q/K/V^T planes)", 6: "out_proj (+ LN residual, row stats)", 7: "linear1 + GELU", 8: "linear2 (K = 1024)"}
else:
    model, _ = make_pair(synth_dip_state_dict(seed=0), 10, dev, guided=True, native_lib=lib, context_len=20, pred_len=40)
    y = to_dev(synth_dip_y(B, 40, 20, seed=1, text_lengths=[24] * B, scale=7.5), dev)
    x = torch.randn(B, 263, 1, 40, device=dev)
    names = {6: "in_proj (norm3 folded -> Q/K/V^T planes)", 7: "out_proj (+ LN residual, row stats)", 8: "cross q-projection (fp32 out)",
             9: "cross out_proj", 10: "linear1 + GELU", 11: "linear2 (K = 1024)"}
t = torch.full((B,), 5, device=dev, dtype=torch.long)
for _ in range(3):
    model(x, t, y=dict(y))
torch.cuda.synchronize()


def read_timeline():
    out = C.c_double()
    vals = np.zeros(4 * 4096)
    for i in range(4 * 4096):
        lib.check(lib.mdm_debug_get(100 + i, C.byref(out)), "debug_get")
        vals[i] = out.value
    return vals.reshape(4096, 4)


def pct(a, q):
    return float(np.percentile(a, q))


for n, name in names.items():
    lib.mdm_debug_set(9, n)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    model(x, t, y=dict(y))
    b.record()
    torch.cuda.synchronize()
    lib.mdm_debug_set(9, -1)
    tl = read_timeline()
    live = tl[:, 0] > 0
    t0max = tl[live, 0].max()
    cur = live & (tl[:, 0] > t0max - 2e6) & (tl[:, 3] >= tl[:, 0])      # this launch's workgroups (older launches' stamps are far behind)
    w = tl[cur]
    base = w[:, 0].min()
    span = w[:, 3].max() - base
    print(f"== launch {n}: {name}: {len(w)} workgroups, first entry -> last store {span:.0f} ticks")
    for lab, v in (("entry offset", w[:, 0] - base), ("entry -> chunk 0 visible", w[:, 1] - w[:, 0]), ("k-loop", w[:, 2] - w[:, 1]),
                   ("epilogue", w[:, 3] - w[:, 2]), ("workgroup lifetime", w[:, 3] - w[:, 0])):
        print(f"   {lab:26s} p10 {pct(v, 10):9.0f}  median {pct(v, 50):9.0f}  p90 {pct(v, 90):9.0f}  max {v.max():9.0f}")
# tick calibration: a busy-wait free estimate from the DiP forward itself is not possible; print the clock the runtime reports
print("ticks: __builtin_readcyclecounter (s_memtime, shader clock)")
