"""Probe: the reference's newest text-to-motion checkpoint shape, `humanml_trans_dec_512_bert-50steps` (README.md:254, trained by
README.md:451: --arch trans_dec --text_encoder_type bert --mask_frames, 50 steps, NO prefix, generate.py's plain p_sample_loop over
196 frames): motions/s at the headline batch and at the CLI's default batch, parity against the oracle at a small batch.
    python lab/probes/transdec50.py [--batch 128] [--reps 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import dip, make_pair, maxabs, orc, synth_dip_state_dict, synth_dip_y, to_dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[128, 6, 1])
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--opts", type=str, default="{}")
a = ap.parse_args()
DEV, T, STEPS, NTOK = "cuda:0", 196, 50, 24
sd = synth_dip_state_dict(seed=0)
rec = {}
model, diffusion = make_pair(sd, STEPS, DEV, guided=True, context_len=0, pred_len=0, mask_frames=True)
model.model.engine_options = json.loads(a.opts)
# parity first (B = 2, ragged frames and prompts, 6 of the 50 steps would not exercise t = 0: run all 50)
B = 2
y = synth_dip_y(B, T, 1, seed=5, text_lengths=[NTOK, 7], lengths=[196, 120], scale=2.5)
y.pop("prefix")
shape = (B, 263, 1, T)
x_T, noises = orc.make_noise(shape, STEPS, 3)
got = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": to_dev(y, DEV)},
                              noise_sequence=[x_T] + [n.contiguous() for n in noises])
want = dip.dip_sample_loop(sd, orc.Tables(orc.named_betas("cosine", STEPS)), shape, y, x_T, noises, context_len=0, cfg=True, mask_frames=True)
rec["parity_B2_50steps"] = maxabs(got.cpu(), want)
print(rec, flush=True)
for B in a.batch:
    g = torch.Generator().manual_seed(B)
    tl = [int(v) for v in torch.randint(6, NTOK + 1, (B,), generator=g)]
    tl[0] = NTOK
    y = synth_dip_y(B, T, 1, seed=7, text_lengths=tl, scale=2.5)
    y.pop("prefix")
    y = to_dev(y, DEV)
    shape = (B, 263, 1, T)
    run = lambda: diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=1)   # noqa: E731
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    rec[f"B{B}"] = {"ms_per_loop": round(1e3 * min(ts), 2), "motions_per_s": round(B / min(ts), 2)}
    print(rec, flush=True)
print(json.dumps(rec))
