"""x3s8 experiment on the MI355X: the DiP forward (B = 3, 64-row tiles) against the upstream reference's fixture, through
whatever library MDM_HIP_LIB names.  Usage: MDM_HIP_LIB=build/libmdm_hip_x3s8.so MDM_X3S_RT=2 python tools/x3s8/check_gpu.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import make_pair, maxabs, synth_dip_state_dict, synth_dip_y, to_dev

g = np.load(os.path.join(ROOT, "tests", "golden", "dip_fwd_B3.npz"))
sd = synth_dip_state_dict(seed=0)
model, _ = make_pair(sd, 10, "cuda:0", guided=True, context_len=20, pred_len=40, mask_frames=False)
y = to_dev(synth_dip_y(3, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), lengths=None), "cuda:0")
x = torch.randn(3, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to("cuda:0")
t = torch.from_numpy(g["t"]).to("cuda:0")
e = [maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"]), maxabs(model.model(x, t, y={**y, "uncond": True}).cpu(), g["out_uncond"]),
     maxabs(model(x, t, y=dict(y)).cpu(), g["out_cfg"])]
print("[x3s8 parity] lib", os.environ.get("MDM_HIP_LIB", "product"), "kinds", os.environ.get("MDM_X3S8_KINDS", "default"),
      "DiP forward B=3 vs reference: %.3e / %.3e / %.3e" % tuple(e), "OK" if e[0] < 2e-5 and e[1] < 2e-5 and e[2] < 5e-5 else "FAIL")
