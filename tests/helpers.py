"""Shared test scaffolding: build the MI355X model/diffusion pair from the synthetic state-dict the golden
fixtures were made with, and run golden loop cases through the product seams."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import mdm_amd  # noqa: E402,F401
from mdm_amd import model_util  # noqa: E402
from mdm_amd.cfg_sampler import ClassifierFreeSampleModel  # noqa: E402
from oracle import mdm_oracle as orc  # noqa: E402
from oracle import dip_oracle as dip  # noqa: E402
from oracle.synth import synth_dip_state_dict, synth_dip_y, synth_state_dict, synth_y  # noqa: E402


def make_pair(sd, steps, device, guided=True, native_lib=None, precision="f16x3", **arg_over):
    """(model, diffusion) exactly as sample/generate.py:85-96 assembles them."""
    dec = any(k.startswith("seqTransDecoder.layers.") for k in sd)
    layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith(("seqTransEncoder.layers.", "seqTransDecoder.layers.")))
    if dec:   # DiP: `--arch trans_dec --text_encoder_type bert --context_len 20 --pred_len 40` (DiP.md)
        arg_over = {"arch": "trans_dec", "text_encoder_type": "bert", "mask_frames": True, **arg_over}   # DiP.md:181 trains with --mask_frames
    d = sd["input_process.poseEmbedding.weight"].shape[0]
    if native_lib is not None:
        # CPU emulator runs: a 512-row positional / timestep table instead of the reference's 5000 (utils/parser_util.py
        # pos_embed_max_len).  mdm_prepare evaluates the TimestepEmbedder for EVERY row of the table -- two [max_len, d] x [d, d] GEMMs
        # -- which the lock-step emulator spends ~7 s on per engine, i.e. a third of the CPU suite's time over its ~200 engines.  The GPU
        # suite binds the full table; tests/test_emu_path.py::test_emulated_cfg_loop_matches_oracle keeps the 5000-row one on the emulator.
        arg_over = {"pos_embed_max_len": 512, **arg_over}
    args = model_util.default_args(diffusion_steps=steps, layers=layers, latent_dim=d, **arg_over)
    model, diffusion = model_util.create_model_and_diffusion(args, _native_lib=native_lib, num_heads=d // 128,
                                                             precision=precision)
    model_util.load_model_wo_clip(model, sd)
    if guided:
        model = ClassifierFreeSampleModel(model)
    model.to(device)
    model.eval()
    return model, diffusion


def golden_loop_inputs(g):
    """Rebuild the inputs of one tests/golden loop case (same recipe as oracle/make_golden.py)."""
    steps, B, T, seed, skip = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"]), int(g["skip"])
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]), scale=float(g["scale"]))
    gi = torch.Generator().manual_seed(seed + 2000)
    init_image = torch.randn(*shape, generator=gi) if bool(g["init"]) else None
    if bool(g["inpaint"]):
        m = torch.zeros(shape, dtype=torch.bool)
        m[:, :4, :, :] = True
        m[..., : T // 4] = True
        y["inpainting_mask"] = m
        y["inpainted_motion"] = torch.randn(*shape, generator=gi)
    x_T, noises = orc.make_noise(shape, steps - skip, seed)
    return dict(steps=steps, shape=shape, y=y, init_image=init_image, skip=skip, x_T=x_T, noises=noises,
                cfg=bool(g["cfg"]), ddim=bool(g["ddim"]), eta=float(g["eta"]))


def run_product_loop(sd, case, device, native_lib=None, dump_steps=None, precision="f16x3"):
    """The golden case through SpacedDiffusion.p_sample_loop / ddim_sample_loop of the product."""
    model, diffusion = make_pair(sd, case["steps"], device, guided=case["cfg"], native_lib=native_lib,
                                 precision=precision)
    seq = [case["x_T"]] + [n.contiguous() for n in case["noises"]]
    kw = dict(clip_denoised=False, model_kwargs={"y": dict(case["y"])}, skip_timesteps=case["skip"],
              init_image=case["init_image"], noise_sequence=seq)
    if case["ddim"]:
        return diffusion.ddim_sample_loop(model, case["shape"], eta=case["eta"], **kw)
    return diffusion.p_sample_loop(model, case["shape"], dump_steps=dump_steps, **kw)


def small_state_dict(seed=0, latent_dim=256, num_layers=2, ff_size=1024):
    """A shallow/narrow configuration for the CPU emulator (same key names; head dim stays 128)."""
    return synth_state_dict(seed=seed, latent_dim=latent_dim, ff_size=ff_size, num_layers=num_layers)


def maxabs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


DIP_CONTEXT, DIP_PRED = 20, 40


def dip_small_state_dict(seed=0, latent_dim=256, num_layers=1, ff_size=1024):
    """A shallow/narrow DiP decoder for the CPU emulator."""
    return synth_dip_state_dict(seed=seed, latent_dim=latent_dim, ff_size=ff_size, num_layers=num_layers)


def to_dev(y, device):
    """model_kwargs['y'] with its tensors (and the (embedding, mask) tuple of DiP) moved to `device`."""
    out = {}
    for k, v in y.items():
        if isinstance(v, tuple):
            out[k] = tuple(t.to(device) for t in v)
        elif torch.is_tensor(v):
            out[k] = v.to(device)
        else:
            out[k] = v
    return out


_MEMO = {}


def memo(key, fn):
    """Oracle results are pure functions of their seeded inputs: tests that run once per GEMM kernel / arithmetic mode
    (tests/conftest.py gemm_path) compute the CPU oracle once (the GPU suite's time is mostly oracle time)."""
    if key not in _MEMO:
        _MEMO[key] = fn()
    return _MEMO[key]
