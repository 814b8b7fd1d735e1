"""TEST INFRASTRUCTURE (build container only: needs /root/reference): INTEGRATION.md section 1, executed.

Runs as a subprocess of tests/test_round2_cpu.py with MDM_HIP_LIB pointing at the CPU emulation of the library:
  1. rebinds the reference's names exactly as INTEGRATION.md's launcher does (utils.model_util.MDM / SpacedDiffusion /
     space_timesteps, diffusion.respace.SpacedDiffusion, utils.sampler_util.ClassifierFreeSampleModel);
  2. builds (model, diffusion) through the REFERENCE's own utils.model_util.create_model_and_diffusion and loads a state
     dict through its load_model_wo_clip;
  3. replays sample/generate.py:93-158's call sequence (wrap -> .to() -> .eval() -> collate -> scale -> text_embed ->
     sample_fn(...) with generate.py's exact keyword set) and checks the samples against the oracle on the same noise;
  4. exercises the hand-over to the reference's own sampler for what has no native path (cond_fn, PLMS, a foreign model).
sample.generate itself cannot be imported offline (clip, moviepy, spacy, datasets: SURVEY.md 8b).  Prints one JSON line."""
import json
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_harness as rh   # noqa: E402
from oracle import mdm_oracle as orc   # noqa: E402
from oracle.synth import synth_state_dict   # noqa: E402

rh._install_stubs()                    # `clip` and `model.rotation2xyz` are not installable offline

# ---- the launcher of INTEGRATION.md section 1, verbatim up to `import sample.generate`
import mdm_amd                                   # noqa: E402,F401
from mdm_amd.mdm import MDM                      # noqa: E402
from mdm_amd.respace import SpacedDiffusion, space_timesteps   # noqa: E402
from mdm_amd.cfg_sampler import ClassifierFreeSampleModel      # noqa: E402

import utils.model_util as model_util            # noqa: E402  (reference module)
import utils.sampler_util as sampler_util        # noqa: E402  (reference module)
import diffusion.respace as respace              # noqa: E402
ref_MDM = model_util.MDM
model_util.MDM = MDM
model_util.SpacedDiffusion = SpacedDiffusion
model_util.space_timesteps = space_timesteps
respace.SpacedDiffusion = SpacedDiffusion
sampler_util.ClassifierFreeSampleModel = ClassifierFreeSampleModel


def maxabs(a, b):
    return float((a.double() - b.double()).abs().max())


def main():
    torch.set_num_threads(2)
    res = {}
    steps, B, T, D, L = 2, 2, 8, 512, 1      # (the schedule needs >= 2 steps; the emulator costs ~20 s per D=512 forward)
    args = SimpleNamespace(dataset="humanml", latent_dim=D, layers=L, arch="trans_enc", emb_trans_dec=False,
                           cond_mask_prob=0.1, text_encoder_type="clip", pos_embed_max_len=5000, mask_frames=True,
                           unconstrained=False, diffusion_steps=steps, noise_schedule="cosine", sigma_small=True,
                           lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0, guidance_param=2.5)
    data = SimpleNamespace(dataset=SimpleNamespace())
    model, diffusion = model_util.create_model_and_diffusion(args, data)          # the reference's factory, our classes
    assert type(model) is MDM and type(diffusion) is SpacedDiffusion
    sd = synth_state_dict(seed=0, latent_dim=D, num_layers=L)
    ck = dict(sd)
    ck["sequence_pos_encoder.pe"] = torch.zeros(5000, 1, D)                       # checkpoints carry them; the loader deletes them
    ck["embed_timestep.sequence_pos_encoder.pe"] = torch.zeros(5000, 1, D)
    model_util.load_model_wo_clip(model, ck)                                      # the reference's loader (:8-15)

    # ---- generate.py:93-158
    sample_fn = diffusion.p_sample_loop
    model = sampler_util.ClassifierFreeSampleModel(model)
    model.to("cpu")
    model.eval()
    motion_shape = (B, model.njoints, model.nfeats, T)
    from data_loaders.tensors import collate                                      # reference, imports as is
    collate_args = [{"inp": torch.zeros(T), "tokens": None, "lengths": n} for n in (T, 5)]
    collate_args = [dict(arg, text=txt) for arg, txt in zip(collate_args, ["a person walks", "a person jumps"])]
    _, model_kwargs = collate(collate_args)
    model_kwargs["y"] = {k: v.to("cpu") if torch.is_tensor(v) else v for k, v in model_kwargs["y"].items()}
    model_kwargs["y"]["scale"] = torch.ones(B) * args.guidance_param
    # generate.py:130-132 caches model.encode_text(y['text']); CLIP is not installable offline, so the cached embedding is
    # synthetic and the 'text' key is dropped
    del model_kwargs["y"]["text"]
    model_kwargs["y"]["text_embed"] = torch.randn(1, B, 512, generator=torch.Generator().manual_seed(5))
    torch.manual_seed(10)                                                         # utils/fixseed.py
    sample = sample_fn(model, motion_shape, clip_denoised=False, model_kwargs=model_kwargs, skip_timesteps=0,
                       init_image=None, progress=True, dump_steps=None, noise=None, const_noise=False)
    assert tuple(sample.shape) == motion_shape and bool(torch.isfinite(sample).all())
    eng = model.model.engine()
    seq = [eng.randn(motion_shape, "cpu", diffusion._seed, 0, k) for k in range(steps + 1)]
    y = model_kwargs["y"]
    tab = orc.Tables(orc.named_betas("cosine", steps))
    want = orc.sample_loop(sd, tab, motion_shape, y, seq[0], seq[1:], cfg=True, num_heads=D // 128)
    res["generate_call_sequence_vs_oracle"] = maxabs(sample, want)

    # ---- no native path -> the reference's own sampler takes over
    x_T, noises = orc.make_noise(motion_shape, steps, 77)
    want_cpu = orc.sample_loop(sd, tab, motion_shape, y, x_T, noises, cfg=True, num_heads=D // 128)
    torch.manual_seed(77)      # the MI355X model driven by the reference's generator: two (emulated) steps, zero gradient
    got = diffusion.p_sample_loop(model, motion_shape, clip_denoised=False, model_kwargs=model_kwargs,
                                  cond_fn=lambda x, t, **kw: torch.zeros_like(x))
    res["cond_fn_handover_vs_oracle"] = maxabs(got, want_cpu)
    # this implementation's noise-stream arguments cannot be honoured by the reference's sampler: refused, not dropped
    for bad in (dict(seed=5), dict(noise_sequence=[x_T] + list(noises))):
        try:
            diffusion.p_sample_loop(model, motion_shape, clip_denoised=False, model_kwargs=model_kwargs,
                                    cond_fn=lambda x, t, **kw: torch.zeros_like(x), **bad)
            raise AssertionError(f"{list(bad)} was silently dropped on a reference-routed call")
        except ValueError:
            pass
    assert type(diffusion._reference("test").base).__module__ == "diffusion.gaussian_diffusion"
    # a foreign model: the reference's own torch MDM through OUR diffusion object
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        foreign = ref_MDM(**rh.mdm_kwargs(latent_dim=D, num_layers=L))
    foreign.load_state_dict(sd, strict=False)
    foreign.eval()
    from model.cfg_sampler import ClassifierFreeSampleModel as RefCFG
    torch.manual_seed(77)
    got = diffusion.p_sample_loop(RefCFG(foreign), motion_shape, clip_denoised=False, model_kwargs=model_kwargs)
    res["foreign_model_handover_vs_oracle"] = maxabs(got, want_cpu)
    out = diffusion.plms_sample_loop(RefCFG(foreign), motion_shape, clip_denoised=False, model_kwargs=model_kwargs)
    res["plms_runs"] = bool(tuple(out.shape) == motion_shape and torch.isfinite(out).all())
    print("DROPIN " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
