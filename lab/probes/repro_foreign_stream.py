"""Companion of tools/repro_dip_groups.py (open issue, profiles/r02e_dip.md): does the PRODUCT library's single-chain DiP window
loop stay bit-reproducible while a foreign stream of the same process keeps dispatching kernels on the device?
    python tools/repro_foreign_stream.py [f16x3|f32] [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from helpers import make_pair, synth_dip_state_dict, synth_dip_y, to_dev
DEV = "cuda:0"
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sdd = synth_dip_state_dict(seed=0)
B = 8
model, diffusion = make_pair(sdd, 10, DEV, guided=True, context_len=20, pred_len=40, precision=prec)
y = to_dev(synth_dip_y(B, 40, 20, seed=2, text_lengths=[4, 11, 25, 8, 1, 17, 9, 30]), DEV)
run = lambda: diffusion.p_sample_loop(model, (B, 263, 1, 40), clip_denoised=False, model_kwargs={"y": y}, seed=7).cpu()
ref = run()
side = torch.cuda.Stream()
bufs = [torch.randn(3840, 512, device=DEV) for _ in range(4)]
w = torch.randn(512, 512, device=DEV)
q = torch.randn(16, 8, 256, 64, device=DEV, dtype=torch.float16)
kinds = ("gemm-sized torch ops", "tiny elementwise ops", "sdpa (LDS-heavy attention kernels)")
if len(sys.argv) > 3:     # a substring selects the foreign load(s)
    kinds = tuple(k for k in kinds if sys.argv[3] in k)
for kind in kinds:
    fails = 0
    for rep in range(reps):
        with torch.cuda.stream(side):
            for i in range(400):
                if kind.startswith("gemm"):
                    bufs[(i + 1) % 4] = torch.nn.functional.layer_norm(bufs[i % 4] @ w, (512,))
                elif kind.startswith("sdpa"):
                    q = torch.nn.functional.scaled_dot_product_attention(q, q, q)
                else:
                    bufs[i % 4].mul_(1.0001).add_(0.1)
        fails += int(not torch.equal(run(), ref))
        torch.cuda.synchronize()
    print(prec, "one chain under", kind, "on a foreign stream: differing window loops", fails, "of", reps)
