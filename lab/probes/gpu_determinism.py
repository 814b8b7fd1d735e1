"""Run-to-run determinism screen of one CFG denoiser forward (model-level; every encoder kernel on the path).
Usage: python tools/gpu_determinism.py [B=128] [reps=8] [layers=8]      (MDM_X3_PIPE=0/1 selects the k-loop: one process each)
Prints, per repetition, whether the output equals the first run's bit for bit, the max-abs difference and which samples differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdm_amd  # noqa: F401
from mdm_amd import model_util
from mdm_amd.cfg_sampler import ClassifierFreeSampleModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
T, dev = 196, "cuda:0"
torch.manual_seed(0)
args = model_util.default_args(diffusion_steps=50, layers=layers)
mdm, _ = model_util.create_model_and_diffusion(args)
model = ClassifierFreeSampleModel(mdm).to(dev).eval()
g = torch.Generator().manual_seed(1)
lengths = torch.tensor([196 - (7 * i) % 150 for i in range(B)])
y = {"mask": (torch.arange(T)[None, :] < lengths[:, None]).view(B, 1, 1, T).to(dev), "lengths": lengths.to(dev),
     "text_embed": torch.randn(1, B, 512, generator=g).to(dev), "scale": torch.full((B,), 2.5, device=dev)}
x = torch.randn(B, 263, 1, T, generator=g).to(dev)
t = torch.full((B,), 25, dtype=torch.long, device=dev)
with torch.no_grad():
    ref = model(x, t, y=dict(y)).clone()
    torch.cuda.synchronize()
    bad = 0
    for r in range(reps):
        out = model(x, t, y=dict(y))
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            d = (out - ref).abs()
            per = d.flatten(1).max(dim=1).values
            idx = torch.nonzero(per > 0).flatten().tolist()
            print(f"  rep {r}: DIFFERS  max-abs {float(d.max()):.3e}  samples {idx[:12]}{'...' if len(idx) > 12 else ''} "
                  f"({len(idx)} of {B}); first frames differing in sample {idx[0]}: "
                  f"{torch.nonzero(d[idx[0], :, 0, :].max(dim=0).values > 0).flatten().tolist()[:16]}")
print(f"PIPE={os.environ.get('MDM_X3_PIPE', '1')} B={B} layers={layers}: {bad} of {reps} repetitions differ from the first run "
      f"(|out| max {float(ref.abs().max()):.3f}, finite {bool(torch.isfinite(ref).all())})")
