"""Where does an selfattn_block_kernel launch (csrc/selfattn_block.h: self MODE 0/1, cross MODE 2) spend its time?  Shader-clock stamps of wave 0 of every workgroup of ONE
selected launch (probe library: SB_STAMP; mdm_debug_set(11, n) / mdm_debug_get(300000 + ...)).  Runs guided DiP forwards at B motions
(2 B sequences of 20 + 40 tokens) and prints the distribution of the seven phases over the workgroups of layer 1's launch.
Usage: python tools/xb_timeline.py [B=32]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from helpers import make_pair, synth_dip_state_dict, synth_dip_y, to_dev
from mdm_amd import _native

lib = _native.load_probe()
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, _ = make_pair(synth_dip_state_dict(seed=0), 10, dev, guided=True, native_lib=lib, context_len=20, pred_len=40)
y = to_dev(synth_dip_y(B, 40, 20, seed=1, text_lengths=[24] * B, scale=7.5), dev)
x = torch.randn(B, 263, 1, 40, device=dev)
t = torch.full((B,), 5, device=dev, dtype=torch.long)
for _ in range(3):
    model(x, t, y=dict(y))
torch.cuda.synchronize()
NW = 1024


def read_timeline():
    out = C.c_double()
    vals = np.zeros(8 * NW)
    for i in range(8 * NW):
        lib.check(lib.mdm_debug_get(300000 + i, C.byref(out)), "debug_get")
        vals[i] = out.value
    return vals.reshape(NW, 8)


def pct(a, q):
    return float(np.percentile(a, q))


LABELS = ["entry -> chunk 0 / vectors / mask (/ K, V^T images of the memory) visible", "contraction (in_proj of the head | q projection) incl. ring drain",
          "fold epilogue -> fragment images (+ barrier)", "attention arithmetic", "plane stores"]
for n, what in ((2, "layer 1 self-attention (MODE 1)"), (3, "layer 1 cross-attention (MODE 2)")):
    lib.mdm_debug_set(11, n)
    model(x, t, y=dict(y))
    torch.cuda.synchronize()
    lib.mdm_debug_set(11, -1)
    tl = read_timeline()
    live = tl[:, 0] > 0
    t0max = tl[live, 0].max()
    cur = live & (tl[:, 0] > t0max - 2e6) & (tl[:, 5] >= tl[:, 0])
    w = tl[cur]
    print(f"== {what}: {len(w)} workgroups; lifetime median {pct(w[:, 5] - w[:, 0], 50):.0f} ticks, p90 {pct(w[:, 5] - w[:, 0], 90):.0f}, "
          f"max {(w[:, 5] - w[:, 0]).max():.0f}; first entry -> last store {w[:, 5].max() - w[:, 0].min():.0f}")
    for i, lab in enumerate(LABELS):
        v = w[:, i + 1] - w[:, i]
        print(f"   {lab:84s} p10 {pct(v, 10):8.0f}  median {pct(v, 50):8.0f}  p90 {pct(v, 90):8.0f}")
print("ticks: __builtin_readcyclecounter (s_memtime: 100 MHz constant clock on gfx9 -- compare phases, or scale by the launch's duration)")
