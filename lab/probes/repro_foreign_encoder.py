"""Companion of tools/repro_foreign_stream.py for the ENCODER path (the headline kernels): does one CFG denoiser forward stay
bit-reproducible while a foreign stream of the same process runs torch scaled_dot_product_attention kernels on the device?
    python tools/repro_foreign_encoder.py [B=32] [reps=40]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdm_amd  # noqa: F401
from mdm_amd import model_util
from mdm_amd.cfg_sampler import ClassifierFreeSampleModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
T, dev = 196, "cuda:0"
torch.manual_seed(0)
args = model_util.default_args(diffusion_steps=50)
mdm, _ = model_util.create_model_and_diffusion(args)
model = ClassifierFreeSampleModel(mdm).to(dev).eval()
g = torch.Generator().manual_seed(1)
lengths = torch.tensor([196 - (7 * i) % 150 for i in range(B)])
y = {"mask": (torch.arange(T)[None, :] < lengths[:, None]).view(B, 1, 1, T).to(dev), "lengths": lengths.to(dev),
     "text_embed": torch.randn(1, B, 512, generator=g).to(dev), "scale": torch.full((B,), 2.5, device=dev)}
x = torch.randn(B, 263, 1, T, generator=g).to(dev)
t = torch.full((B,), 25, dtype=torch.long, device=dev)
side = torch.cuda.Stream()
q = torch.randn(16, 8, 256, 64, device=dev, dtype=torch.float16)
with torch.no_grad():
    ref = model(x, t, y=dict(y)).clone()
    torch.cuda.synchronize()
    for kind in ("idle device", "sdpa (LDS-heavy attention kernels) on a foreign stream"):
        bad = 0
        for r in range(reps):
            if kind.startswith("sdpa"):
                with torch.cuda.stream(side):
                    for i in range(300):
                        q = torch.nn.functional.scaled_dot_product_attention(q, q, q)
            outs = [model(x, t, y=dict(y)) for _ in range(3)]
            torch.cuda.synchronize()
            bad += int(any(not torch.equal(o, ref) for o in outs))
        print(f"encoder forward B={B} under {kind}: repetitions (3 forwards each) with a differing forward: {bad} of {reps}")
